/*
 * gaot3d_hip.h -- C ABI of libgaot3d_hip.so: the MI355X (gfx950) kernels behind the GAOT-3D
 * forward/backward hot path.  Plain pointers and sizes only (no torch types).  All pointers are
 * DEVICE pointers borrowed for the duration of the call unless marked "host"; every entry point
 * enqueues on the given hipStream_t and returns without synchronising.  Return value: 0 = ok,
 * otherwise an error code; gaot_last_error() gives the message (thread local).
 *
 * Each entry point names the reference interface it stands in for (paths relative to the
 * reference repo Shizheng-Wen/GAOT-3D); the Python host side (gaot_3d_amd/model) mirrors the
 * reference's src/model operator API on top of these.
 */
#ifndef GAOT3D_HIP_H
#define GAOT3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAOT_OK 0
#define GAOT_ERR_ARG 1
#define GAOT_ERR_LAUNCH 2
#define GAOT_ERR_UNSUPPORTED 3

#define GAOT_ABI_VERSION 1
#define GAOT_MAX_MLP_LAYERS 5

typedef void* gaot_stream_t; /* hipStream_t */

int gaot_abi_version(void);
const char* gaot_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Neighbour lists.  Replaces the implicit "group edges by index" of the scatter calls
 * (reference src/model/layers/utils/scatter_native.py:4-31; torch_scatter.scatter) with an
 * explicit row-sorted list.  edge_index is the reference's [2,E] tensor (row 0 = source, row 1 =
 * query; int32 on disk, int64 after collate: src/trainer/stat.py:191,208, integral_transform.py
 * :114-115), contiguous.  Sorting is stable (original edge order inside a row).
 *   rowptr[num_rows+1], perm[E] (sorted position -> original edge id),
 *   key_sorted[E] (= sorted row index), other_sorted[E] (= the other endpoint).
 * ------------------------------------------------------------------------------------------- */
size_t gaot_csr_workspace_bytes(int64_t num_edges, int64_t num_rows);
int gaot_csr_build(const void* edge_index, int index_is_i64, int64_t num_edges, int sort_row, int64_t num_rows,
                   int32_t* rowptr, int32_t* perm, int32_t* key_sorted, int32_t* other_sorted, void* workspace,
                   size_t workspace_bytes, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GNO kernel integral transform, transform_type="linear", reduction="mean", no attention
 * (reference IntegralTransform.forward, src/model/layers/integral_transform.py:80-175, with the
 * LinearChannelMLP of src/model/layers/mlp.py:327-335):
 *   out[q,:] = mean_{e=(s->q)} MLP([y_pos[s], x_pos[q]]) * f_y[s,:]      (empty row -> 0)
 * MLP = 6 -> hidden x n_hidden -> channels, erf-GELU between layers, weights in nn.Linear layout
 * [out][in].  Edges are given sorted by query (gaot_csr_build with sort_row=1).
 * Supported shapes: coord dim 3, hidden in {32,64,128}, channels == 32, 1 <= n_hidden <= 4.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int n_hidden;
    int hidden;
    int channels;
    const float* weight[GAOT_MAX_MLP_LAYERS]; /* layer l: [out_l][in_l] */
    const float* bias[GAOT_MAX_MLP_LAYERS];
} gaot_mlp_t;

typedef struct {
    float* weight[GAOT_MAX_MLP_LAYERS];
    float* bias[GAOT_MAX_MLP_LAYERS];
} gaot_mlp_grad_t;

size_t gaot_gno_fwd_workspace_bytes(int64_t num_edges, int channels);
int gaot_gno_fwd(const gaot_mlp_t* mlp /* host */, const float* y_pos, const float* x_pos, const float* f_y,
                 const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_dst, int64_t num_edges,
                 int64_t num_queries, float* out /* [num_queries, channels] */, void* workspace,
                 size_t workspace_bytes, gaot_stream_t stream);

/* Backward of the above (autograd of the reference ops): grad wrt f_y and wrt every MLP
 * parameter; none wrt coordinates.  Edges are given sorted by SOURCE (gaot_csr_build with
 * sort_row=0) so that grad_f_y is again an atomics-free segmented sum; rowptr_dst (from the
 * by-query list) supplies the mean's 1/deg.  grad outputs are overwritten, not accumulated. */
size_t gaot_gno_bwd_workspace_bytes(const gaot_mlp_t* mlp /* host */, int64_t num_edges);
int gaot_gno_bwd(const gaot_mlp_t* mlp /* host */, const float* y_pos, const float* x_pos, const float* f_y,
                 const float* grad_out /* [num_queries, channels] */, const int32_t* rowptr_dst,
                 const int32_t* src_sorted /* by source */, const int32_t* dst_sorted /* by source */,
                 const int32_t* rowptr_src, int64_t num_edges, int64_t num_sources, int64_t num_queries,
                 float* grad_f_y /* [num_sources, channels] */, const gaot_mlp_grad_t* grads /* host */,
                 void* workspace, size_t workspace_bytes, gaot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAOT3D_HIP_H */
