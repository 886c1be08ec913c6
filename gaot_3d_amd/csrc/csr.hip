// Neighbour-list (CSR) builder: turns the PyG-style bipartite ``edge_index [2,E]`` the reference
// feeds to IntegralTransform (reference src/model/layers/integral_transform.py:114-115: row 0 =
// source index, row 1 = query index) into a list sorted by one of the two rows, so that the
// torch_scatter-style segmented reductions (reference scatter_native.py:21-31) become contiguous
// wavefront segmented sums with no atomics in the data path.
//
// Stable counting sort, all on device, no host sync:
//   1. histogram of keys (int atomics, L2)         2. exclusive scan -> rowptr
//   3. scatter edge ids through per-row cursors     4. per-row rank sort of the edge ids
// Step 4 makes the order inside a row = original edge order, so every later floating-point
// segmented sum is bit-reproducible run to run.
//
// Fast path: neighbour searches emit their lists grouped by query, so one of the two orders a
// bipartite graph needs is usually sorted already.  A first pass checks that on the device (no host
// sync, graph-capturable); if the keys are non-decreasing the list IS its own stable sort: one
// streaming pass writes rowptr / perm / the copies, and steps 1-4 return at their first instruction.
#include "common.h"

namespace {

// *unsorted != 0 when some key is smaller than its predecessor
template <typename IDX>
__global__ void k_check_sorted(const IDX* __restrict__ keys, int64_t E, int* __restrict__ unsorted) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < E; i += stride) bad |= (i > 0) && (keys[i] < keys[i - 1]);
    if (__any(bad) && (threadIdx.x & 63) == 0) *unsorted = 1;   // benign race: every writer stores 1
}

// sorted keys: the input order is the stable sort.  rowptr[r] = first edge with key >= r.
template <typename IDX>
__global__ void k_sorted_build(const IDX* __restrict__ keys, const IDX* __restrict__ other, int64_t E, int64_t Q,
                               const int* __restrict__ unsorted, int* __restrict__ rowptr, int* __restrict__ perm,
                               int* __restrict__ key_sorted, int* __restrict__ other_sorted) {
    if (*unsorted) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) {
        const int k = (int)keys[i];
        perm[i] = (int)i;
        key_sorted[i] = k;
        other_sorted[i] = (int)other[i];
        const int kprev = (i > 0) ? (int)keys[i - 1] : -1;
        for (int r = kprev + 1; r <= k; ++r) rowptr[r] = (int)i;      // rows (kprev, k] start here (empty ones too)
        if (i == E - 1)
            for (int64_t r = (int64_t)k + 1; r <= Q; ++r) rowptr[r] = (int)E;
    }
}

template <typename IDX>
__global__ void k_hist(const IDX* __restrict__ keys, int64_t E, int* __restrict__ counts, const int* __restrict__ unsorted) {
    if (!*unsorted) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) atomicAdd(&counts[(int)keys[i]], 1);
}

// ---- 3-phase exclusive scan over n ints (n up to ~2^31) -------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;  // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__global__ void k_scan_reduce(const int* __restrict__ in, int64_t n, int* __restrict__ block_sums) {
    __shared__ int red[SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int64_t i = base + (int64_t)threadIdx.x * SCAN_ITEMS + j;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += red[w];
        block_sums[blockIdx.x] = t;
    }
}

// single block: exclusive scan of block_sums in place
__global__ void k_scan_blocksums(int* __restrict__ block_sums, int nb) {
    __shared__ int carry;
    __shared__ int wsum[SCAN_BLOCK / 64];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += SCAN_BLOCK) {
        int i = base + threadIdx.x;
        int v = (i < nb) ? block_sums[i] : 0;
        int incl = v;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int c = carry;
        if (i < nb) block_sums[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry = c + woff + incl;
        __syncthreads();
    }
}

// out[i] = exclusive prefix; also copies to cursor[i]; writes out[n] = total when i==n-1
__global__ void k_scan_apply(const int* __restrict__ in, int64_t n, const int* __restrict__ block_offs,
                             int* __restrict__ out, int* __restrict__ cursor, const int* __restrict__ unsorted) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    if (!*unsorted) return;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0;
        s += v[j];
    }
    int incl = s;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int off = block_offs[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    int run = off + incl - s;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) {
            out[base + j] = run;
            cursor[base + j] = run;
        }
        run += v[j];
        if (base + j == n - 1) out[n] = run;
    }
}

template <typename IDX>
__global__ void k_fill(const IDX* __restrict__ keys, int64_t E, int* __restrict__ cursor, int* __restrict__ tmp,
                       const int* __restrict__ unsorted) {
    if (!*unsorted) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) {
        int p = atomicAdd(&cursor[(int)keys[i]], 1);
        tmp[p] = (int)i;
    }
}

constexpr int HEAVY_DEG = 128;     // rows longer than this go to the block-wide LDS sort
constexpr int BITONIC_MAX = 8192;  // ints in LDS (32 KB)

template <typename IDX>
__device__ __forceinline__ void emit_sorted(int p, int id, int64_t row, const IDX* __restrict__ other, int* perm,
                                            int* key_sorted, int* other_sorted) {
    perm[p] = id;
    key_sorted[p] = (int)row;
    other_sorted[p] = (int)other[id];
}

// Light rows: G lanes cooperate on one row with a rank sort of the (unique) edge ids -- O(n^2/G) L1 hits,
// fine for short rows.  Rows longer than HEAVY_DEG are queued for k_row_sort_heavy instead.  G in {8, 64}.
template <typename IDX, int G>
__global__ void k_row_sort(const IDX* __restrict__ keys, const IDX* __restrict__ other, int64_t Q,
                           const int* __restrict__ rowptr, const int* __restrict__ tmp, int* __restrict__ perm,
                           int* __restrict__ key_sorted, int* __restrict__ other_sorted, int* __restrict__ heavy_count,
                           int* __restrict__ heavy_list, const int* __restrict__ unsorted) {
    if (!*unsorted) return;
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int gl = threadIdx.x % G;
    if (gid >= Q) return;
    const int b = rowptr[gid], e = rowptr[gid + 1];
    const int n = e - b;
    if (n > HEAVY_DEG) {
        if (gl == 0) heavy_list[atomicAdd(heavy_count, 1)] = (int)gid;
        return;
    }
    for (int i = gl; i < n; i += G) {
        const int id = tmp[b + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (tmp[b + j] < id) ? 1 : 0;
        emit_sorted<IDX>(b + rank, id, gid, other, perm, key_sorted, other_sorted);
    }
}

// Heavy rows: one workgroup per row (persistent over the queue).  n <= BITONIC_MAX: bitonic sort of the ids
// in LDS (n log^2 n); longer rows: chunked rank sort against LDS-resident chunks.  Which block sorts which
// row does not matter: the result (ascending edge id inside the row) is unique.
template <typename IDX>
__global__ __launch_bounds__(256) void k_row_sort_heavy(const IDX* __restrict__ other, const int* __restrict__ rowptr,
                                                        const int* __restrict__ tmp, int* __restrict__ perm,
                                                        int* __restrict__ key_sorted, int* __restrict__ other_sorted,
                                                        const int* __restrict__ heavy_count,
                                                        const int* __restrict__ heavy_list) {
    __shared__ int sh[BITONIC_MAX];
    const int nheavy = *heavy_count;
    for (int w = blockIdx.x; w < nheavy; w += gridDim.x) {
        const int row = heavy_list[w];
        const int b = rowptr[row], n = rowptr[row + 1] - b;
        __syncthreads();
        if (n <= BITONIC_MAX) {
            int np = 1;
            while (np < n) np <<= 1;
            for (int i = threadIdx.x; i < np; i += 256) sh[i] = (i < n) ? tmp[b + i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= np; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < np; i += 256) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const int a = sh[i], c = sh[ixj];
                            const bool up = (i & k) == 0;
                            if ((a > c) == up) { sh[i] = c; sh[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = threadIdx.x; i < n; i += 256) emit_sorted<IDX>(b + i, sh[i], row, other, perm, key_sorted, other_sorted);
        } else {
            for (int t0 = 0; t0 < n; t0 += 256) {       // 256 targets per pass, one per thread
                const int ti = t0 + threadIdx.x;
                const int id = (ti < n) ? tmp[b + ti] : 0;
                int rank = 0;
                for (int c0 = 0; c0 < n; c0 += BITONIC_MAX) {
                    const int cn = min(BITONIC_MAX, n - c0);
                    __syncthreads();
                    for (int i = threadIdx.x; i < cn; i += 256) sh[i] = tmp[b + c0 + i];
                    __syncthreads();
                    if (ti < n)
                        for (int i = 0; i < cn; ++i) rank += (sh[i] < id) ? 1 : 0;
                }
                if (ti < n) emit_sorted<IDX>(b + rank, id, row, other, perm, key_sorted, other_sorted);
            }
        }
    }
}

template <typename IDX>
int csr_build_t(const IDX* edge_index, int64_t E, int sort_row, int64_t Q, int32_t* rowptr, int32_t* perm,
                int32_t* key_sorted, int32_t* other_sorted, void* ws, hipStream_t st) {
    const IDX* keys = edge_index + (sort_row ? E : 0);
    const IDX* other = edge_index + (sort_row ? 0 : E);
    const int64_t n = Q;  // counts has Q entries (+1 slot for the total)
    const int nb = (int)ceil_div(n, SCAN_TILE);
    int* counts = (int*)ws;                 // [Q+1]
    int* cursor = counts + (Q + 1);         // [Q+1]
    int* bsum = cursor + (Q + 1);           // [nb+1]
    int* tmp = bsum + (nb + 1);             // [E]
    int* heavy_count = tmp + E;             // [1] (+3 pad)
    int* heavy_list = heavy_count + 4;      // [<= E / HEAVY_DEG + 1]
    int* unsorted = heavy_count + 1;        // cleared together with heavy_count
    (void)hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(Q + 1), st);
    (void)hipMemsetAsync(heavy_count, 0, sizeof(int) * 4, st);
    if (E == 0) {
        (void)hipMemsetAsync(rowptr, 0, sizeof(int) * (size_t)(Q + 1), st);
        return GAOT_OK;
    }
    const int tb = 256;
    const int gb = (int)std::min<int64_t>(ceil_div(E, tb), 256 * 16);
    hipLaunchKernelGGL((k_check_sorted<IDX>), dim3(gb), dim3(tb), 0, st, keys, E, unsorted);
    hipLaunchKernelGGL((k_sorted_build<IDX>), dim3(gb), dim3(tb), 0, st, keys, other, E, Q, unsorted, rowptr, perm,
                       key_sorted, other_sorted);
    hipLaunchKernelGGL((k_hist<IDX>), dim3(gb), dim3(tb), 0, st, keys, E, counts, unsorted);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, st, counts, n, bsum);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_BLOCK), 0, st, bsum, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, st, counts, n, bsum, rowptr, cursor, unsorted);
    hipLaunchKernelGGL((k_fill<IDX>), dim3(gb), dim3(tb), 0, st, keys, E, cursor, tmp, unsorted);
    if (E / (Q > 0 ? Q : 1) >= 16) {
        const int64_t threads = Q * 64;
        hipLaunchKernelGGL((k_row_sort<IDX, 64>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, keys, other,
                           Q, rowptr, tmp, perm, key_sorted, other_sorted, heavy_count, heavy_list, unsorted);
    } else {
        const int64_t threads = Q * 8;
        hipLaunchKernelGGL((k_row_sort<IDX, 8>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, keys, other,
                           Q, rowptr, tmp, perm, key_sorted, other_sorted, heavy_count, heavy_list, unsorted);
    }
    hipLaunchKernelGGL((k_row_sort_heavy<IDX>), dim3(1024), dim3(256), 0, st, other, rowptr, tmp, perm, key_sorted,
                       other_sorted, heavy_count, heavy_list);
    return GAOT_OK;
}

}  // namespace

extern "C" size_t gaot_csr_workspace_bytes(int64_t num_edges, int64_t num_rows) {
    const int64_t nb = ceil_div(num_rows, SCAN_TILE);
    return sizeof(int) * (size_t)(2 * (num_rows + 1) + (nb + 1) + num_edges + 4 + num_edges / HEAVY_DEG + 1) + 64;
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
