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

#define GAOT_ABI_VERSION 11
#define GAOT_MAX_MLP_LAYERS 5

typedef void* gaot_stream_t; /* hipStream_t */

int gaot_abi_version(void);
const char* gaot_last_error(void);
/* kernel launches issued by this library in this process since the last reset (measurement aid for bench.py: the reference
 * has no counterpart; its step is a few thousand ATen launches, src/trainer/optimizers.py:272-275) */
int64_t gaot_launch_count(int reset);

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
                 int64_t num_queries, float* out /* [num_queries, channels] */,
                 int precision /* 0: exact-fp32 MFMA; 1: bf16 MFMA for the hidden/last layers, fp32 accumulate */,
                 void* workspace, size_t workspace_bytes, gaot_stream_t stream);

/* Backward of the above (autograd of the reference ops): grad wrt f_y and wrt every MLP
 * parameter; none wrt coordinates.  Edges are given sorted by SOURCE (gaot_csr_build with
 * sort_row=0) so that grad_f_y is again an atomics-free segmented sum; rowptr_dst (from the
 * by-query list) supplies the mean's 1/deg.  grad outputs are overwritten, not accumulated. */
size_t gaot_gno_bwd_workspace_bytes(const gaot_mlp_t* mlp /* host */, int64_t num_edges, int64_t num_queries);
int gaot_gno_bwd(const gaot_mlp_t* mlp /* host */, const float* y_pos, const float* x_pos, const float* f_y,
                 const float* grad_out /* [num_queries, channels] */, const int32_t* rowptr_dst,
                 const int32_t* src_sorted /* by source */, const int32_t* dst_sorted /* by source */,
                 const int32_t* rowptr_src, int64_t num_edges, int64_t num_sources, int64_t num_queries,
                 float* grad_f_y /* [num_sources, channels] */, const gaot_mlp_grad_t* grads /* host */,
                 int precision, void* workspace, size_t workspace_bytes, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Statistical geometric-embedding features (reference GeometricEmbedding.
 * _compute_statistical_features_pyg, src/model/layers/geoembed.py:99-182): z-scored
 * [N_i, D_avg, D_var, centroid-query (3), covariance eigenvalues desc (3)] per query row, from the
 * by-query neighbour list.  features: [num_queries, 9].
 * ------------------------------------------------------------------------------------------- */
size_t gaot_geoembed_stats_workspace_bytes(void);
/* One sweep over the neighbour list: moments[row][12] = {N, sum d, sum d^2, sum u (3), sum u u^T (6)}, u = x - q
 * (fp64) are plain sums over the row's edges -> (point-sharded sample: SUM all-reduce them across the ranks that hold the
 * edges, then) gaot_geoembed_from_moments finishes (variance, centred covariance, eigenvalues, column z-score) on every rank. */
int gaot_geoembed_moments(const float* source_pos, const float* query_pos, const int32_t* rowptr_dst,
                          const int32_t* src_sorted, int64_t num_queries, double* moments, gaot_stream_t stream);
int gaot_geoembed_from_moments(const double* moments, int64_t num_queries, float* features, void* workspace,
                               size_t workspace_bytes, gaot_stream_t stream);
/* the same features when the QUERIES are spread over several ranks (decoder side of a point-sharded sample: a rank
 * owns every edge of its queries, only the column z-score of geoembed.py:177-180 runs over all queries):
 * raw = un-normalised local features + their 18 column sums / sums of squares (fp64); after the caller's SUM
 * all-reduce of colsums, finalize applies the z-score over num_queries_total rows. */
int gaot_geoembed_raw(const float* source_pos, const float* query_pos, const int32_t* rowptr_dst, const int32_t* src_sorted,
                      int64_t num_queries, float* features, double* colsums, void* workspace, size_t workspace_bytes,
                      gaot_stream_t stream);
int gaot_geoembed_finalize(float* features, int64_t num_queries, const double* colsums, int64_t num_queries_total,
                           void* workspace, size_t workspace_bytes, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense GEMM  C[m][n] = act(sum_k A(m,k) B(k,n) + bias[n]) + residual[m][n]   (row-major, fp32 I/O)
 * stands in for ATen addmm/mm behind nn.Linear and its autograd (reference mlp.py:327-335,
 * attn.py:104-106,129,156; gaot_3d.py:205).  a_trans: A(m,k) = A[k*lda+m]; b_trans: B(k,n) =
 * B[n*ldb+k] (nn.Linear weight layout).  act: 0 none, 1 GELU(erf), 2 ReLU, 3 SiLU.  preact
 * (optional, ldc) receives the value before the activation.  precision 0 = fp32 MFMA (exact fp32
 * products), 1 = bf16 operands / fp32 accumulate.  Long reductions with few output tiles (weight
 * gradients) are split over K with a fixed-order second pass (workspace).
 * ------------------------------------------------------------------------------------------- */
size_t gaot_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gaot_gemm(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
              int64_t ldc, int a_trans, int b_trans, const float* bias, int act, const float* residual, int64_t ldr,
              float* preact, int precision, void* workspace, size_t workspace_bytes, gaot_stream_t stream);
/* gaot_gemm with operands / result that already ARE bf16 in memory (a_bf16 / b_bf16 / c_bf16 != 0; leading dimensions
 * in elements): the FFN intermediates of the bf16 path ([rows, 2F] = w1 x | w3 x, silu(a)*g, and their gradients,
 * reference attn.py:156) are written once as bf16 by their producer and never exist as fp32 in HBM.  precision must
 * be 1 and N > 64; rows must be 16-byte aligned and K a multiple of 8; a bf16 result excludes split-K.  Supported
 * any combination of bf16 A, B and C. */
int gaot_gemm_ex(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                 int64_t ldc, int a_trans, int b_trans, int a_bf16, int b_bf16, int c_bf16, const float* bias, int act,
                 const float* residual, int64_t ldr, float* preact, int precision, void* workspace,
                 size_t workspace_bytes, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention: softmax(Q K^T * scale) V per head, no mask, optional dropout (reference
 * GroupQueryFlashAttention.forward, src/model/layers/attn.py:110-127 -> F.scaled_dot_product_
 * attention(dropout_p = atten_dropout when training, attn.py:122-127)) and its autograd.  q/k/v/o are [B*S, heads*32] views with row strides ld* (floats),
 * so the fused QKV projection output can be addressed in place.  lse/delta: [B, H, S] scratch
 * kept from forward / filled by backward (phase 1 of gaot_attn_bwd; phases may be issued one by one).  head_dim must be 32.  HKV < H = grouped-query heads
 * (k = k.repeat_interleave(H/HKV)).
 * Dropout (dropout_p > 0): O = (P .* keep / (1-p)) V with keep(seed, b, h, q, k) a counter-based Bernoulli(1-p)
 * mask (csrc/attn_dropout.h; p realised to 1/65536) that forward and backward regenerate from *dropout_seed, a
 * DEVICE pointer to one 64-bit word read when the kernel runs (so a captured hipGraph sees the value of each
 * replay); the backward must be given the value the forward saw.  torch's own Philox mask is not reproducible
 * outside torch; gaot_attn_dropout_mask materialises this one (keep[b][h][q][k], 1 byte each) for checks.
 * head0 / heads_total (0, 0 = the launch holds all heads): the launch's H heads are heads head0 .. head0+H-1 of heads_total --
 * one rank's slice in a head- / sequence-parallel step (no reference counterpart: stat.py:431-436 is sample-level DDP only);
 * the mask is keyed by the GLOBAL head index b * heads_total + head0 + h, so all ranks share ONE seed word and a head draws
 * the same mask wherever it runs.
 * ------------------------------------------------------------------------------------------- */
int gaot_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int64_t ldq, int64_t ldk,
                  int64_t ldv, int64_t ldo, int B, int S, int H, int HKV, int head_dim, float scale, float dropout_p,
                  const unsigned long long* dropout_seed, int head0, int heads_total, int precision, gaot_stream_t stream);
int gaot_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                  float* delta, float* dq, float* dk, float* dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int S, int H, int HKV, int head_dim,
                  float scale, float dropout_p, const unsigned long long* dropout_seed, int head0, int heads_total,
                  int precision, int phase_mask /* 1 delta | 2 dK,dV | 4 dQ ; 7 = all */, gaot_stream_t stream);
/* fp32 mode: dK, dV and dQ from ONE pass over the (query tile, key block) pairs (the two-pass form above recomputes S and dP for dQ:
 * 7 S^2 d products per layer, 5 here) -- dQ leaves as fp32 slab partials [B][H][ceil(S/256)][S][32] in `scratch`
 * (gaot_attn_bwd_fused_f32_scratch_bytes(B, S, H)), summed in slab order by the reduction launched behind it: no atomics,
 * bit-reproducible.  run_delta != 0: phase 1 of gaot_attn_bwd first.  Same arguments and results as gaot_attn_bwd(phase_mask 7). */
int64_t gaot_attn_bwd_fused_f32_scratch_bytes(int B, int S, int H);
int gaot_attn_bwd_fused_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                            float* delta, float* dq, float* dk, float* dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                            int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int S, int H, int HKV, int head_dim,
                            float scale, float dropout_p, const unsigned long long* dropout_seed, int head0, int heads_total,
                            int run_delta, void* scratch, size_t scratch_bytes, gaot_stream_t stream);
int gaot_attn_dropout_mask(const unsigned long long* dropout_seed, float dropout_p, int B, int H, int S,
                           unsigned char* keep, gaot_stream_t stream);
/* the seed stream: *out = *state; *state += stride -- the word one dropout call uses, and the advance, in one launch */
int gaot_dropout_seed_next(unsigned long long* state, unsigned long long stride, unsigned long long* out,
                           gaot_stream_t stream);
/* the words of the next n calls in one launch (ABI 11): out[i] = *state + i * stride, *state += n * stride */
int gaot_dropout_seed_block(unsigned long long* state, unsigned long long stride, int n, unsigned long long* out, gaot_stream_t stream);

/* bf16 matrix-core path of the same operator (precision 1).  qkv is the fused fp32 projection
 * [B*S][(H+2*HKV)*32] (q | k | v column blocks).  Forward first writes a bf16 image of it (RoPE applied when
 * rope_freqs != NULL, q pre-scaled) into qkv_image (gaot_attn_bf16_image_bytes: the image, then the forward's own
 * scratch -- the partial maxima of |k|^2 it bounds the scores with, one flag per workgroup, the key-range parts of
 * launches with few heads), which the backward re-uses;
 * backward writes the fp32 gradient w.r.t. the projection output into dqkv: with rope_freqs (the forward's) the dq / dk
 * tiles are rotated back in the kernels' epilogues; with NULL the gradient is w.r.t. the ROTATED q|k|v.
 * do_image: scratch of gaot_attn_bwd_bf16_scratch_bytes (bf16 dO image; for launches with few heads also the
 * per-range partial gradients).  When (S/128)*H*B is below two workgroups per CU the streamed range (keys for forward
 * and dQ, queries for dK/dV) is split over blockIdx.y and the parts are combined in a fixed order (forward: by their
 * log-sum-exp; backward: summed at the end of phase 4, so phases 2 and 4 must both be issued).
 * phase_mask: 1 = dO image + delta from the fp32 d_o, 2 = dK/dV, 4 = dQ; 16 + 32 (instead of 2 and 4) = fused dK/dV/dQ pass + its slab reduction; 8 (instead of 1) = do_image ALREADY holds the bf16
 * dO (it arrived as bf16 from the sequence-parallel exchange; d_o may be NULL): delta only. */
/* The fused q|k|v projection written straight as that image (reference attn.py:104-109: q_proj / k_proj / v_proj + rotary
 * embedding of q and k; stands in for gaot_gemm_ex followed by the preparation pass of gaot_attn_fwd_bf16, so the fp32
 * projection never exists in HBM): x [rows][256] bf16, w [(H+2*HKV)*32][256] bf16 (the co-located q/k/v weights), image
 * as sized by gaot_attn_bf16_image_bytes.  rope_table = gaot_rope_table's [S][16][2] (cos, sin) for the layer's RoPE
 * frequencies, or NULL; qscale = scale * log2(e) (what the kernels expect folded into q).  Afterwards call
 * gaot_attn_fwd_bf16 with qkv = NULL. */
int gaot_rope_table(const float* freqs, int S, int half_dim, float* table, gaot_stream_t stream);
int gaot_qkv_image(const void* x_bf16, const void* w_bf16, void* image, int64_t rows, int64_t lda, int64_t ldw, int S, int H,
                   int HKV, const float* rope_table, float qscale, gaot_stream_t stream);
/* The same projection for a SEQUENCE-PARALLEL step (one sample's token rows split over `world` ranks, heads split inside
 * attention; no reference counterpart -- src/trainer/stat.py:431-436 knows sample-level DDP only): x holds this rank's rows
 * (positions pos0 .. pos0+rows-1), `packed` receives `world` blocks [rows][(H+2*HKV)/world*32], block j = q | k | v of rank j's
 * heads = the send buffer of the all-to-all; what a rank receives is its attention image ([world*rows][...], its heads).
 * rope_table: gaot_rope_table for the FULL sequence. */
int gaot_qkv_image_packed(const void* x_bf16, const void* w_bf16, void* packed, int64_t rows, int64_t lda, int64_t ldw,
                          int64_t pos0, int H, int HKV, const float* rope_table, float qscale, int world, gaot_stream_t stream);
/* rows layout [rows][ld] (nseg <= 3 column segments, each `world` groups of width[s] columns from col0[s]; group j belongs to
 * rank j) <-> blocks layout [world][rows][sum width] (the all_to_all_single buffers of that step); dtypes 0 = fp32, 1 = bf16
 * on either side (the exchange travels as bf16 in bf16 mode). */
int gaot_pack_heads(void* rows_buf, void* blocks_buf, int64_t rows, int ld, int world, int nseg, const int* col0,
                    const int* width, int rows_dtype, int blocks_dtype, int to_blocks, gaot_stream_t stream);
size_t gaot_attn_bf16_image_bytes(int B, int S, int H, int HKV);
size_t gaot_attn_bwd_bf16_scratch_bytes(int B, int S, int H, int HKV);
/* 1 when phases 16 + 32 of gaot_attn_bwd_bf16 (dK, dV and dQ from ONE pass over the score tiles, workgroups of 512 keys,
 * bf16 dQ slab partials in the scratch; fixed-order slab reduction) can be used instead of phases 2 + 4: ceil(S/512)*HKV*B >= 128 */
int gaot_attn_bwd_bf16_fused_eligible(int B, int S, int H, int HKV);
int gaot_attn_fwd_bf16(const float* qkv, const float* rope_freqs, void* qkv_image, float* o, float* lse, int B, int S,
                       int H, int HKV, int head_dim, float scale, float dropout_p,
                       const unsigned long long* dropout_seed, int head0, int heads_total, gaot_stream_t stream);
int gaot_attn_bwd_bf16(const void* qkv_image, const float* o, const float* d_o, const float* lse, void* do_image,
                       float* delta, float* dqkv, const float* rope_freqs, int B, int S, int H, int HKV, int head_dim,
                       float scale, float dropout_p, const unsigned long long* dropout_seed, int head0, int heads_total,
                       int phase_mask, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row / element kernels (HBM-bound).
 *   rmsnorm : y = x * rsqrt(mean(x^2) + eps) * w         (attn.py:174-178); rstd[rows] kept for bwd
 *   rope    : in-place 1-D rotary embedding over the flattened token index (attn.py:118-120;
 *             rotary_embedding_torch: interleaved pairs, angle = (row % seq_len) * freqs[i]);
 *             inverse=1 applies the transpose rotation (backward)
 *   swiglu  : u = silu(a) * g on a fused [rows][2F] buffer (attn.py:156) and its backward
 *   act_bwd : dz = dh * act'(z)      axpy : out = a + alpha * b[i % period]
 *   patchify: [B,D,H,W,C] <-> [B,S,P^3 C] token layout (gaot_3d.py:199-202, 218-220)
 *   mse     : nn.MSELoss mean reduction (src/trainer/base.py:56) and d(loss)/d(pred) * (*grad_loss)
 *   colsum  : out[n] = sum_m x[m][n]  (bias gradients)
 * ------------------------------------------------------------------------------------------- */
int gaot_rmsnorm_fwd(const float* x, const float* weight, float* y, float* rstd, void* y_bf16 /* NULL, or the same rows
                     rounded to bf16 for the GEMM that consumes them */, int64_t rows, int dim, float eps,
                     gaot_stream_t stream);
size_t gaot_rmsnorm_bwd_workspace_bytes(int64_t rows, int dim);
/* dx_add (may be NULL): a gradient that reaches x through another consumer -- the block's residual
 * `x + attn(norm(x))` (attn.py:226) -- added into dx by the same pass instead of a separate accumulation kernel */
int gaot_rmsnorm_bwd(const float* x, const float* weight, const float* dy, const float* rstd, const float* dx_add,
                     float* dx, float* dweight, int64_t rows, int dim, void* workspace, size_t workspace_bytes,
                     gaot_stream_t stream);
/* the same with a SECOND gradient reaching x through a third consumer (ABI 11: the U-ViT long-range skip, reference attn.py:282-288:
 * an encoder block's output feeds the next block and the mirrored decoder block): dx = (dx + dx_add) + dx_add2 */
int gaot_rmsnorm_bwd2(const float* x, const float* weight, const float* dy, const float* rstd, const float* dx_add,
                      const float* dx_add2, float* dx, float* dweight, int64_t rows, int dim, void* workspace, size_t workspace_bytes,
                      gaot_stream_t stream);
size_t gaot_colsum_workspace_bytes(int64_t M, int64_t N);
int gaot_colsum(const float* x, int64_t M, int64_t N, int64_t ld, float* out, void* workspace, size_t workspace_bytes,
                gaot_stream_t stream);
/* Deferred completion of fixed-order reductions (ABI 10).  The weight gradients of a backward pass (split-K products dy^T x of the
 * nn.Linear sites, reference attn.py:104-106,127,146-156; RMSNorm weights, attn.py:205-230; bias column sums) are read by nobody
 * before the optimizer step, so their final "sum the partials" passes -- ~90 launches of ~5 us in a configs[1] step -- can be
 * handed to ONE launch at the end of the backward pass:
 *   gaot_gemm_ex_partials : gaot_gemm_ex (no epilogue, dense C) that leaves the split-K partials [splits][M*N] in `workspace` when
 *                           its plan splits K (*splits_out > 1; the caller keeps the workspace alive); *splits_out == 1: C is complete
 *   gaot_rmsnorm_bwd with dweight == NULL, gaot_colsum with out == NULL: the gaot_rmsnorm_bwd_parts(rows) x dim /
 *                           gaot_colsum_parts(M) x N partial rows stay in the workspace (lanes = 32)
 *   gaot_reduce_multi     : out_j[i] = sum_p part_j[p*n_j + i] for every descriptor, `lanes` part-lanes summing every lanes-th
 *                           partial and the lane sums added in lane order -- the order of the in-call passes, bit-identical results */
typedef struct {
    const float* part;   /* [parts][n] */
    float* out;          /* [n] */
    int64_t n;
    int32_t parts;
    int32_t lanes;       /* 4, 16 (what gaot_gemm_ex_partials reports) or 32 */
} gaot_reduce_desc_t;
int gaot_gemm_ex_partials(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldc, int a_trans, int b_trans, int a_bf16, int b_bf16, int precision, void* workspace,
                          size_t workspace_bytes, int* splits_out, int* lanes_out, gaot_stream_t stream);
int64_t gaot_rmsnorm_bwd_parts(int64_t rows);
int64_t gaot_colsum_parts(int64_t M);
int gaot_reduce_multi(const gaot_reduce_desc_t* descs, int count, gaot_stream_t stream);
int gaot_rope(float* x, int64_t rows, int64_t ld, int col0, int nheads, int head_dim, int seq_len, const float* freqs,
              int inverse, gaot_stream_t stream);
int gaot_swiglu_fwd(const float* ag, float* u, int64_t rows, int F, gaot_stream_t stream);
int gaot_swiglu_bwd(const float* ag, const float* du, float* dag, int64_t rows, int F, gaot_stream_t stream);
/* fp32 -> bf16 (round to nearest even) copy of a weight matrix: the B operand of the bf16 GEMMs of one step is
 * rounded once instead of once per workgroup that streams it (gaot_gemm_ex, b_bf16) */
int gaot_cast_bf16(const float* src, void* dst, int64_t n, gaot_stream_t stream);
/* the same for many tensors in one launch (all bf16 weight copies of the Transformer, once per forward) */
typedef struct {
    const float* src;
    void* dst;        /* numel bf16 */
    int64_t numel;
} gaot_cast_tensor_t;
int gaot_cast_bf16_multi(const gaot_cast_tensor_t* tensors, int num_tensors, gaot_stream_t stream);
/* the same with a transpose: fp32 [rows[i]][cols[i]] -> bf16 [cols[i]][rows[i]] (numel = rows * cols): the weights of the FFN
 * whose input-gradient GEMMs run as x W^T on the transposed copy (gaot_gemm_ex with K = 256 / N = 256, reference attn.py:156) */
int gaot_cast_bf16_transpose_multi(const gaot_cast_tensor_t* tensors, const int* rows, const int* cols, int num_tensors,
                                   gaot_stream_t stream);
/* the same with every buffer bf16 in memory (fp32 arithmetic); F % 8 == 0, 16-byte aligned buffers */
int gaot_swiglu_fwd_bf16(const void* ag, void* u, int64_t rows, int F, gaot_stream_t stream);
/* the first half of the SwiGLU FFN in ONE launch (reference attn.py:155-156: silu(w1 x) * w3 x): x [rows][256] bf16, w13 = the
 * co-located [w1; w3] weights ([2F][256] bf16) -> ag = w1 x | w3 x (bf16 [rows][2F], kept for the backward) and u = silu(a) g
 * (bf16 [rows][F]); same values as gaot_gemm_ex (bf16 result) followed by gaot_swiglu_fwd_bf16 */
int gaot_ffn_w13_swiglu(const void* x_bf16, const void* w13_bf16, void* ag, void* u, int64_t rows, int64_t lda, int64_t ldw, int F,
                        gaot_stream_t stream);
int gaot_swiglu_bwd_bf16(const void* ag, const void* du, void* dag, int64_t rows, int F, gaot_stream_t stream);
/* the input gradient of the FFN's second projection with the SwiGLU backward in its epilogue (ABI 10; reference attn.py:155-157):
 * dy [rows][256] bf16, w2t = W2^T ([F][256] bf16), ag as saved by gaot_ffn_w13_swiglu -> dag = d(a) | d(g) (bf16 [rows][2F]); same
 * values as gaot_gemm_ex (du = dy W2, bf16 result) followed by gaot_swiglu_bwd_bf16, du never written.  F % 64 == 0 */
int gaot_ffn_w2_bwd_swiglu(const void* dy_bf16, const void* w2t_bf16, const void* ag, void* dag, int64_t rows, int64_t lda,
                           int64_t ldw, int F, gaot_stream_t stream);
/* ---- the SwiGLU FFN as ONE launch per direction over 64-row blocks (ABI 11; csrc/ffn_fused.hip; reference
 * src/model/layers/attn.py:146-157 FFN.forward and :226-229 the residual around it; d_model = 256, F % 128 == 0) ----
 * The weights are streamed from L2 in MFMA fragment order: gaot_ffn_pack turns the fp32 co-located [w1; w3] ([2F][256]) and w2
 * ([256][F]) into the packed bf16 images (gaot_ffn_packed_bytes(F, with_backward) bytes, 16-byte aligned; with_backward adds the
 * images the backward kernel reads) -- one launch per FFN per optimizer step, in place of the bf16 casts / transposes. */
int64_t gaot_ffn_packed_bytes(int F, int with_backward);
int gaot_ffn_pack(const float* w13, const float* w2, int F, void* packed, int with_backward, gaot_stream_t stream);
/* the same for every FFN of a Transformer in ONE launch (all of width F) */
typedef struct {
    const float* w13; /* [2F][256] */
    const float* w2;  /* [256][F] */
    void* packed;     /* gaot_ffn_packed_bytes(F, with_backward) bytes, 16-byte aligned */
} gaot_ffn_pack_t;
int gaot_ffn_pack_multi(const gaot_ffn_pack_t* items, int num, int F, int with_backward, gaot_stream_t stream);
/* y = w2(silu(w1 x) * w3 x) + residual: x [rows][256] bf16, residual fp32 [rows][ldr] or NULL, y fp32 [rows][256]; ag (bf16
 * [rows][2F] = w1 x | w3 x) and u (bf16 [rows][F]) are written for the backward when non-NULL (both or neither).  Bit-identical to
 * gaot_ffn_w13_swiglu followed by gaot_gemm_ex(u, w2, residual). */
int gaot_ffn_fwd(const void* x_bf16, const void* packed, const float* residual, int64_t ldr, float* y, void* ag, void* u, int64_t rows,
                 int F, gaot_stream_t stream);
/* RMSNorm + FFN + residual of a Transformer block in one launch (attn.py:227-229: h = ffn_norm(h); h + ffn(h) -- the residual is the
 * NORMALISED h): y = n + w2(silu(w1 n) * w3 n), n = RMSNorm(h; norm_weight, eps); yb = bf16(n) and rstd are written for the backward
 * (gaot_ffn_bwd on yb, then gaot_rmsnorm_bwd on its dx).  Values of gaot_rmsnorm_fwd followed by gaot_ffn_fwd. */
int gaot_norm_ffn_fwd(const float* h, int64_t ldh, const float* norm_weight, float eps, const void* packed, float* y, void* yb,
                      float* rstd, int64_t rows, int F, gaot_stream_t stream);
/* the whole tail of a Transformer block in one launch (attn.py:127, 226-229): h = x + o_proj(attn_out); n = ffn_norm(h);
 * y = n + w2(silu(w1 n) * w3 n).  gaot_block_pack_multi = gaot_ffn_pack_multi with the backward images plus the fragment image of
 * o_proj.weight ([256][256]); gaot_block_packed_bytes(F) bytes per block.  h (fp32 [rows][256]), yb = bf16(n) and rstd are written
 * for the backward (gaot_ffn_bwd on yb, gaot_rmsnorm_bwd on its dx, then the o_proj products on dh). */
typedef struct {
    const float* w13; /* [2F][256] */
    const float* w2;  /* [256][F] */
    const float* wo;  /* [256][256] */
    void* packed;     /* gaot_block_packed_bytes(F) bytes, 16-byte aligned */
} gaot_block_pack_t;
int64_t gaot_block_packed_bytes(int F);
int gaot_block_pack_multi(const gaot_block_pack_t* items, int num, int F, gaot_stream_t stream);
int gaot_block_tail_fwd(const float* attn_out, int64_t ldo, const float* x, int64_t ldx, const float* norm_weight, float eps,
                        const void* packed, float* h, float* y, void* yb, float* rstd, int64_t rows, int F, gaot_stream_t stream);
/* the head of a Transformer block in one launch (attn.py:104-109, 118-120, 226): q | k | v projections of attn_norm(x) written as the
 * attention kernels' bf16 image (RoPE, q pre-scaled) -- gaot_rmsnorm_fwd + gaot_qkv_image without the fp32 / bf16 round trip of the
 * normalised rows; yb = bf16(attn_norm(x)) and rstd are written for the backward.  gaot_qkv_pack_multi: the co-located fp32 weights
 * ([N][256], N = (H + 2 HKV) * 32 a multiple of 256) of all blocks as fragment images (with_backward: + the transposed image). */
typedef struct {
    const float* w; /* [N][256] */
    void* packed;   /* gaot_qkv_packed_bytes(N, with_backward) bytes, 16-byte aligned */
} gaot_qkv_pack_t;
int64_t gaot_qkv_packed_bytes(int64_t N, int with_backward);
int gaot_qkv_pack_multi(const gaot_qkv_pack_t* items, int num, int64_t N, int with_backward, gaot_stream_t stream);
int gaot_norm_qkv_image(const float* x, int64_t ldx, const float* norm_weight, float eps, const void* packed, void* image, void* yb,
                        float* rstd, int64_t rows, int S, int H, int HKV, const float* rope_table, float qscale, gaot_stream_t stream);
/* gaot_norm_qkv_image with the decoder block's skip projection in front (attn.py:222-225: x = skip_proj(cat([x, skip]))): xa, xb fp32
 * [rows][256]; gaot_skip_pack_multi packs skip_proj.weight ([256][512], gaot_skip_packed_bytes() per block); x_out = the projected rows */
int64_t gaot_skip_packed_bytes(void);
int gaot_skip_pack_multi(const gaot_qkv_pack_t* items, int num, gaot_stream_t stream);
int gaot_cat_norm_qkv_image(const float* xa, int64_t ldxa, const float* xb, int64_t ldxb, const void* skip_packed, const float* skip_bias,
                            float* x_out, const float* norm_weight, float eps, const void* packed, void* image, void* yb, float* rstd,
                            int64_t rows, int S, int H, int HKV, const float* rope_table, float qscale, gaot_stream_t stream);
/* RMSNorm backward as the epilogue of the product that forms d(norm(x)) (ABI 11; reference attn.py:167-178 autograd):
 *  gaot_ffn_bwd_norm  = gaot_ffn_bwd + gaot_rmsnorm_bwd(ffn_norm): dh instead of d(norm(h)); h, rstd as saved by the forward
 *  gaot_qkv_bwd_norm  = gaot_gemm_ex (d(norm x) = dqkv Wqkv, packed WITH the backward image) + gaot_rmsnorm_bwd2(attn_norm): dx, with the
 *                       residual's (dres) and the skip tap's (dtap) gradients as addends (either may be NULL)
 * Both leave the norm weight's gradient as gaot_norm_bwd_parts(rows) partial rows of 256 (one per 64-row block) for gaot_reduce_multi. */
int64_t gaot_norm_bwd_parts(int64_t rows);
int gaot_ffn_bwd_norm(const void* yb, const float* dy, const void* packed, const float* h, int64_t ldh, const float* norm_weight,
                      const float* rstd, void* dag, void* u, void* dyb, float* dh, float* dw_part, int64_t rows, int F, gaot_stream_t stream);
int gaot_qkv_bwd_norm(const float* dqkv, int64_t N, const void* packed, const float* x, int64_t ldx, const float* norm_weight,
                      const float* rstd, const float* dres, const float* dtap, float* dx, float* dw_part, int64_t rows, gaot_stream_t stream);
/* gaot_qkv_bwd_norm for a decoder block (x = skip_proj(cat([xa, xb]))): also dxa = dx Ws[:, :256] and dxb = dx Ws[:, 256:] from the dx
 * rows on chip (same != 0: xa is xb -- their sum in dxa); skip_packed = gaot_skip_pack_multi's image (it carries Ws^T as well) */
int gaot_qkv_bwd_norm_cat(const float* dqkv, int64_t N, const void* packed, const float* x, int64_t ldx, const float* norm_weight,
                          const float* rstd, const float* dres, const void* skip_packed, float* dx, float* dxa, float* dxb, int same,
                          float* dw_part, int64_t rows, gaot_stream_t stream);
/* the o_proj backward's input gradient d_o = dh Wo written straight as the flash backward's operands (attn.py:122-127 autograd): the
 * bf16 dO image [rows][256] and delta[rows / S][8][S] = sum over a head's 32 columns of d_o * attn_out -- stands in for gaot_gemm_ex and
 * phase 1 of gaot_attn_bwd_bf16 (call it with phases 16 | 32 only).  packed: a block image of gaot_block_pack_multi. */
int gaot_oproj_bwd_image(const float* dh, const float* attn_out, const void* packed, int F, void* do_image, float* delta, int64_t rows,
                         int S, gaot_stream_t stream);
/* the first half of the backward for a forward that saved nothing (ag = u = NULL above): a | g recomputed from x, du = dy W2, the
 * SwiGLU derivative -> dag = d(a) | d(g) (bf16 [rows][2F]), u = silu(a) g (bf16 [rows][F]), dyb = bf16(dy) ([rows][256], optional):
 * the operands of the dx = dag W13, dW13 = dag^T x and dW2 = dyb^T u products.  packed: gaot_ffn_pack WITH the backward images.
 * Values of gaot_ffn_w13_swiglu + gaot_gemm_ex (du, bf16 result) + gaot_swiglu_bwd_bf16 + gaot_cast_bf16. */
int gaot_ffn_bwd_dag(const void* x_bf16, const float* dy, const void* packed, void* dag, void* u, void* dyb, int64_t rows, int F,
                     gaot_stream_t stream);
/* the same with the input gradient in the launch: dx fp32 [rows][256] = dag W13 (+ dy when add_dy: the block's residual is the FFN's
 * own input, attn.py:229), accumulated chunk by chunk from the dag chunk on chip -- stands in for gaot_ffn_bwd_dag followed by
 * gaot_gemm_ex(dag, W13^T, residual = dy); dag and u are still written (the dW13 / dW2 products read them) */
int gaot_ffn_bwd(const void* x_bf16, const float* dy, const void* packed, void* dag, void* u, void* dyb, float* dx, int add_dy,
                 int64_t rows, int F, gaot_stream_t stream);
/* Activations outside the GEMM epilogue's none / gelu / relu / silu: the rest of the reference's `activation_fn(name)`
 * surface (src/model/layers/mlp.py:27-35: any F.<name>, torch's default parameters).  act ids: 0 none, 1 gelu (erf), 2 relu,
 * 3 silu, 4 tanh, 5 leaky_relu, 6 elu, 7 sigmoid, 8 softplus, 9 selu, 10 relu6, 11 hardswish, 12 mish, 13 gelu (tanh form).
 * gaot_act_fwd: h = act(z); gaot_act_bwd: dz = dh * act'(z) (every id). */
int gaot_act_fwd(const float* z, float* h, int64_t n, int act, gaot_stream_t stream);
int gaot_act_bwd(const float* z, const float* dh, float* dz, int64_t n, int act, gaot_stream_t stream);
int gaot_axpy(const float* a, const float* b, float alpha, float* out, int64_t n, int64_t period, gaot_stream_t stream);
/* dst = src as a float4 grid-stride copy: the streaming-copy rate bench.py reports beside the 8 TB/s spec figure
 * (SURVEY 8d "also report vs a measured streaming-copy peak"; no reference counterpart) */
int gaot_stream_copy(const void* src, void* dst, int64_t bytes, gaot_stream_t stream);
/* the same copy in a chosen form (measurement only): 0 = grid-stride plain; 1..3 = a contiguous 64-KiB chunk per workgroup,
 * 8 loads in flight per thread, non-temporal loads + stores / plain loads + non-temporal stores / plain; 4..6 = the same with
 * 256-KiB chunks and 4 loads in flight; 7..9 = non-temporal with 16 / 8 / 4 loads in flight and 128-KiB / 1-MiB / 16-KiB chunks.
 * gaot_stream_copy runs the fastest form measured on MI355X. */
int gaot_stream_copy_ex(const void* src, void* dst, int64_t bytes, int variant, gaot_stream_t stream);
int gaot_patchify(const float* src, float* dst, int B, int D, int H, int W, int P, int C, int to_tokens,
                  gaot_stream_t stream);
size_t gaot_mse_workspace_bytes(void);
int gaot_mse_fwd(const float* pred, const float* target, int64_t n, float* loss, void* workspace, size_t workspace_bytes,
                 gaot_stream_t stream);
int gaot_mse_bwd(const float* pred, const float* target, int64_t n, const float* grad_loss, float* dpred,
                 gaot_stream_t stream);

/* Multi-scale mix (reference magno.py:590-594 / 784-788): out = sum_s softmax(logits)_s * x_s with per-node
 * logits [n, num_scales]; xs / dxs are HOST arrays of device pointers ([n, 32] each); weights [n, num_scales]
 * keeps the softmax for the backward. */
int gaot_scale_mix_fwd(const float* const* xs, int num_scales, const float* logits, float* out, float* weights,
                       int64_t n, int channels, gaot_stream_t stream);
int gaot_scale_mix_bwd(const float* const* xs, int num_scales, const float* weights, const float* dout,
                       float* const* dxs, float* dlogits, int64_t n, int channels, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW step: stands in for torch.optim.AdamW(params, lr, weight_decay).step()
 * (reference src/trainer/optimizers.py:210, 272-275; defaults betas (0.9, 0.999), eps 1e-8).  The
 * tensor table is a HOST array of device pointers; lr and the step counter are DEVICE scalars
 * (step is advanced by this call), so the call can be captured into a graph.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} gaot_adamw_tensor_t;
int gaot_adamw_step(const gaot_adamw_tensor_t* tensors, int num_tensors, const float* lr, float* step, double beta1,
                    double beta2, double eps, double weight_decay, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Neighbour-graph construction on the device (reference get_neighbor_strategy, magno.py:116-295;
 * torch_cluster knn / radius on the CPU in the reference).  The latent tokens are a regular
 * D x H x W grid (gaot_3d.py:35-46, stat.py:238-252): gaot_grid_t bounds the search window,
 * token_pos [D*H*W, 3] supplies the coordinates the distances are measured to.
 *   gaot_knn_grid          pyg_knn(x=latent, y=phys, k): out_idx[i*k + j] = j-th nearest token of
 *                          point i, ordered by (distance, token index); any 1 <= k <= number of tokens (torch_cluster
 *                          takes any k, magno.py:183-189): register lists up to 64, passes of 64 over all tokens beyond
 *   gaot_radius_grid_*     pyg_radius(x=latent, y=phys, r, max_num_neighbors=cap): tokens with d <= r
 *                          of every point, ascending token index, at most cap; count pass, exclusive
 *                          scan (gaot_exclusive_scan_i32 -> offsets[n+1]), fill pass
 *   gaot_segment_cap_flags on a list sorted by key (gaot_csr_build): 1 for the first cap elements of
 *                          every segment (the per-centre cap of pyg_radius(x=phys, y=latent))
 *   gaot_unique_pair_flags / gaot_compact_pairs   coalesce (magno.py:219-220) after two stable sorts
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dims[3];
    float lo[3];
    float hi[3];
} gaot_grid_t;
int gaot_knn_grid(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos, int k,
                  int32_t* out_idx, gaot_stream_t stream);
int gaot_radius_grid_count(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos,
                           float radius, int cap, int32_t* counts, gaot_stream_t stream);
int gaot_radius_grid_fill(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos,
                          float radius, int cap, const int32_t* offsets, int32_t* out_point, int32_t* out_token,
                          gaot_stream_t stream);
/* The same searches against token coordinates that are NOT a regular grid (custom `tokens_pos`; the reference's
 * get_neighbor_strategy takes any latent coordinates, magno.py:116-124): every point scans all tokens, staged through LDS
 * -- O(N M) on the device, no N x M distance matrix.  Same order and tie rules as the grid forms. */
int gaot_knn_brute(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens, int k, int32_t* out_idx,
                   gaot_stream_t stream);
int gaot_radius_brute_count(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens, float radius,
                            int cap, int32_t* counts, gaot_stream_t stream);
int gaot_radius_brute_fill(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens, float radius,
                           int cap, const int32_t* offsets, int32_t* out_point, int32_t* out_token, gaot_stream_t stream);
size_t gaot_exclusive_scan_workspace_bytes(int64_t n);
int gaot_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* workspace, size_t workspace_bytes,
                            gaot_stream_t stream);
int gaot_segment_cap_flags(const int32_t* rowptr, const int32_t* key_sorted, int64_t n, int cap, int32_t* flags,
                           gaot_stream_t stream);
int gaot_unique_pair_flags(const int32_t* a, const int32_t* b, int64_t n, int32_t* flags, gaot_stream_t stream);
/* neighbour sampling (reference magno.py:297-371, apply_neighbor_sampling): 'ratio' keeps every edge with probability
 * keep_prob (torch_geometric dropout_edge); 'max_neighbors' keeps a uniformly random `cap` of the edges of every
 * query that has more (randperm[:cap] in the reference).  The draws are a counter-based hash of (*seed, position),
 * seed = DEVICE pointer to one 64-bit word; compact with gaot_exclusive_scan_i32 + gaot_compact_pairs. */
int gaot_random_keep_flags(const unsigned long long* seed, int64_t n, double keep_prob, int32_t* flags,
                           gaot_stream_t stream);
/* nn.Dropout of the reference's channel MLPs (src/model/layers/mlp.py:268-272, 318-322, training mode):
 * out[i] = keep(i) ? x[i] / (1 - p) : 0, keep(i) = hash(seed word, i) >= round(p 2^32); call it on the gradient with
 * the same seed for the backward.  (torch's own Philox draw cannot be reproduced: the draw is "parity unpinned".) */
int gaot_dropout(const float* x, const uint64_t* seed, double p, int64_t n, float* out, gaot_stream_t stream);
int gaot_segment_random_cap_flags(const unsigned long long* seed, const int32_t* rowptr, const int32_t* key_sorted,
                                  int64_t n, int cap, int32_t* flags, gaot_stream_t stream);
int gaot_compact_pairs(const int32_t* a, const int32_t* b, const int32_t* flags, const int32_t* offsets, int64_t n,
                       int32_t* out_a, int32_t* out_b, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-node two-layer MLP  out = W2 gelu(W1 x + b1) + b2  and its autograd: the decoder's
 * projection C -> 256 -> out (reference magno.py:793-797 -> LinearChannelMLP / ChannelMLP with GELU,
 * mlp.py:227-335), bf16 operands / fp32 accumulation.  x [rows, 32], w1 [hidden, 32], w2 [out, hidden]
 * row-major fp32; hidden in {64, 128, 256}, out in 1..4.  The [rows, hidden] activations never reach
 * HBM; backward recomputes them and reduces the weight gradients in a fixed order.  d_b2 is the
 * column sum of d_out (gaot_colsum).
 * ------------------------------------------------------------------------------------------- */
int gaot_mlp2_fwd(const float* x, int64_t num_rows, int in_dim, int hidden, int out_dim, const float* w1, const float* b1,
                  const float* w2, const float* b2, float* out, gaot_stream_t stream);
size_t gaot_mlp2_bwd_workspace_bytes(int hidden, int out_dim);
int gaot_mlp2_bwd(const float* x, int64_t num_rows, int in_dim, int hidden, int out_dim, const float* w1, const float* b1,
                  const float* w2, const float* d_out, float* d_x, float* d_w1, float* d_b1, float* d_w2, void* workspace,
                  size_t workspace_bytes, gaot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * General (unfused) per-edge operators: the glue around the GEMM-run edge MLP for the GNO variants the
 * fused kernels do not cover -- IntegralTransform transform_type "nonlinear" / "nonlinear_kernelonly" and
 * the segment-softmax attention weights (reference integral_transform.py:68-78, 126-160), other kernel-MLP
 * widths, PointNet GeometricEmbedding (geoembed.py:184-222) -- and their autograd.  Per-edge tensors
 * are row-major fp32 in the dst-sorted edge order of gaot_csr_build (a segment = a contiguous row range);
 * a reduction over the other endpoint walks that endpoint's CSR through `map` (position in its order ->
 * position in the dst order).  Fixed-order reductions, no atomics.
 *   gather_rows        out[e] = table[idx[e]]                       (backward: segment_reduce sum over idx's CSR)
 *   segment_reduce     out[r] = sum | mean | max | min over the row's edges (mode 0 | 1 | 2 | 3; empty row -> 0, like
 *                      scatter_native.py:4-54); max / min also return the arg-extremum edge per (row, channel)
 *   segment_reduce_bwd d_vals from d_out (sum / mean: broadcast [/ degree]; max / min: routed to the arg edge)
 *   segment_softmax    w = exp(s - max_seg) / max(sum_seg, FLT_MIN)   and   ds = w * (dw - sum_seg(w dw))
 *   edge_coords        mode 0: [y[src], x[dst]] (6)   1: y[src] - x[dst] (3)   2: cos(x[dst], y[src]) (1)
 *   mul / mul_rowsum   a .* b with b per element or per row;  row-wise sum of a .* b
 * ------------------------------------------------------------------------------------------- */
int gaot_gather_rows(const float* table, int64_t ld, const int32_t* idx, int64_t num_edges, int channels, float* out,
                     int64_t ld_out, gaot_stream_t stream);
int gaot_segment_reduce(const float* vals, int64_t ld_vals, const int32_t* rowptr, const int32_t* map, int64_t num_rows,
                        int channels, int mode, float* out, int32_t* argmax, gaot_stream_t stream);
int gaot_segment_reduce_bwd(const float* d_out, const int32_t* key, const int32_t* rowptr, const int32_t* argmax,
                            int64_t num_rows, int64_t num_edges, int channels, int mode, float* d_vals,
                            gaot_stream_t stream);
int gaot_segment_softmax_fwd(const float* scores, const int32_t* rowptr, int64_t num_rows, float* w, gaot_stream_t stream);
int gaot_segment_softmax_bwd(const float* w, const float* dw, const int32_t* rowptr, int64_t num_rows, float* ds,
                             gaot_stream_t stream);
int gaot_edge_coords(const float* y_pos, const float* x_pos, const int32_t* src, const int32_t* dst, int64_t num_edges,
                     int mode, float* out, int64_t ld_out, gaot_stream_t stream);
int gaot_mul(const float* a, const float* b, int64_t rows, int channels, int b_is_row_scalar, float* out,
             gaot_stream_t stream);
int gaot_mul_rowsum(const float* a, const float* b, int64_t rows, int channels, float* out, gaot_stream_t stream);
/* time-conditioned norm (reference ConditionedNorm.forward, src/model/layers/mlp.py:112-128):
 * out[r][c] = x[r][c] * (1 + scale_minus_one[c]) + bias[c]  for the rows of one batch element; bias may be NULL */
int gaot_affine_cols(const float* x, const float* scale_minus_one, const float* bias, int64_t rows, int channels,
                     float* out, gaot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAOT3D_HIP_H */
