// Neighbour-list (CSR) builder: turns the PyG-style bipartite ``edge_index [2,E]`` the reference
// feeds to IntegralTransform (reference src/model/layers/integral_transform.py:114-115: row 0 =
// source index, row 1 = query index) into a list sorted by one of the two rows, so that the
// torch_scatter-style segmented reductions (reference scatter_native.py:21-31) become contiguous
// wavefront segmented sums with no atomics in the data path.
//
// Stable sort by key, all on device, no host sync, no global atomics: LSD radix sort of (key, edge id)
// with 6-8 bit digits (2-3 passes for 64 K-16 M rows), then one streaming pass that writes rowptr (from
// the key changes), the permutation and the sorted copies.  The order inside a row = original edge
// order, so every later floating-point segmented sum is bit-reproducible run to run.
//
// Fast path: neighbour searches emit their lists grouped by query, so one of the two orders a
// bipartite graph needs is usually sorted already.  A first pass checks that on the device (no host
// sync, graph-capturable); if the keys are non-decreasing the list IS its own stable sort: one
// streaming pass writes rowptr / perm / the copies, and the sort kernels return at their first instruction.
#include "csr_impl.h"

extern "C" size_t gaot_csr_workspace_bytes(int64_t num_edges, int64_t num_rows) {
    (void)num_rows;
    const int64_t nblk = ceil_div(num_edges, RS_TILE);
    const int64_t tn = 256 * nblk;
    const int64_t nbt = ceil_div(tn, SCAN_TILE);
    return sizeof(int) * (size_t)(4 + 4 * num_edges + 2 * (tn + 1) + nbt + 1) + 64;
}

extern "C" int gaot_csr_build(const void* edge_index, int index_is_i64, int64_t num_edges, int sort_row,
                              int64_t num_rows, int32_t* rowptr, int32_t* perm, int32_t* key_sorted,
                              int32_t* other_sorted, void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_edges >= 0 && num_rows >= 0, "negative size");
    GAOT_CHECK_ARG(num_edges < (int64_t)1 << 31 && num_rows < ((int64_t)1 << 31) - 1, "sizes must fit int32");
    GAOT_CHECK_ARG(sort_row == 0 || sort_row == 1, "sort_row must be 0 or 1");
    GAOT_CHECK_ARG(rowptr && workspace, "null pointer");
    GAOT_CHECK_ARG(num_edges == 0 || (edge_index && perm && key_sorted && other_sorted), "null pointer");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_csr_workspace_bytes(num_edges, num_rows), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc = index_is_i64 ? csr_build_t<int64_t>((const int64_t*)edge_index, num_edges, sort_row, num_rows, rowptr,
                                                 perm, key_sorted, other_sorted, workspace, st)
                          : csr_build_t<int32_t>((const int32_t*)edge_index, num_edges, sort_row, num_rows, rowptr,
                                                 perm, key_sorted, other_sorted, workspace, st);
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
