// Shared integer machinery of the neighbour-list code (included by csr.hip and graph.hip): 3-phase exclusive scan,
// stable LSD radix sort of (key, id) pairs, sortedness check and the streaming rowptr/permutation emitters.
// Everything lives in an anonymous namespace: each including translation unit gets its own copy.
#pragma once
#include "common.h"

namespace {

__global__ void k_zero_flags(int* __restrict__ p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}

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

// out[i] = exclusive prefix; writes out[n] = total when i==n-1
__global__ void k_scan_apply(const int* __restrict__ in, int64_t n, const int* __restrict__ block_offs,
                             int* __restrict__ out) {
    __shared__ int wsum[SCAN_BLOCK / 64];
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
        }
        run += v[j];
        if (base + j == n - 1) out[n] = run;
    }
}

// ---- stable LSD radix sort of (key, edge id), 6-8 bit digits, 4096 keys per workgroup ----------------------
// Per pass: per-workgroup digit histogram -> exclusive scan of the [digit][workgroup] table -> scatter with a
// stable rank (ballot matching inside a wave round, per-wave running counts, waves in order).  No global atomics:
// the result is the unique stable order, so every later fp32 segmented sum is bit-reproducible.
constexpr int RS_TILE = 4096;
constexpr int RS_ROUNDS = RS_TILE / 256;   // keys per thread

template <typename KIN>
__global__ __launch_bounds__(256) void k_rs_hist(const KIN* __restrict__ keys, int64_t E, int shift, int nd, int nblk,
                                                 int* __restrict__ table, const int* __restrict__ unsorted) {
    __shared__ int h[256];
    if (!*unsorted) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < E) atomicAdd(&h[((int)keys[i] >> shift) & (nd - 1)], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < nd) table[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

template <typename KIN>
__global__ __launch_bounds__(256) void k_rs_scatter(const KIN* __restrict__ keys_in, const int* __restrict__ ids_in,
                                                    int64_t E, int shift, int nbits, int nblk,
                                                    const int* __restrict__ table, int* __restrict__ keys_out,
                                                    int* __restrict__ ids_out, const int* __restrict__ unsorted) {
    __shared__ int wcnt[4][256];
    if (!*unsorted) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nd = 1 << nbits;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (RS_ROUNDS * 64);
    for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    int key[RS_ROUNDS], rnk[RS_ROUNDS];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const int64_t idx = base + j * 64 + lane;
        const bool valid = idx < E;
        const int k = valid ? (int)keys_in[idx] : 0;
        const int d = (k >> shift) & (nd - 1);
        unsigned long long m = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1;
            const unsigned long long bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        const int r = __popcll(m & lt);
        const int before = wcnt[wave][d];
        rnk[j] = before + r;
        key[j] = k;
        if (valid && r == 0) wcnt[wave][d] = before + __popcll(m);   // the group's first lane; same-wave LDS ops stay in order
    }
    __syncthreads();
    if ((int)threadIdx.x < nd) {
        int run = table[(int64_t)threadIdx.x * nblk + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = wcnt[w][threadIdx.x];
            wcnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const int64_t idx = base + j * 64 + lane;
        if (idx < E) {
            const int pos = wcnt[wave][(key[j] >> shift) & (nd - 1)] + rnk[j];
            keys_out[pos] = key[j];
            ids_out[pos] = ids_in ? ids_in[idx] : (int)idx;
        }
    }
}

// sorted (key, id) pairs -> rowptr / perm / key / other (general path; mirrors k_sorted_build)
template <typename IDX>
__global__ void k_emit_sorted(const int* __restrict__ keys, const int* __restrict__ ids, const IDX* __restrict__ other,
                              int64_t E, int64_t Q, const int* __restrict__ unsorted, int* __restrict__ rowptr,
                              int* __restrict__ perm, int* __restrict__ key_sorted, int* __restrict__ other_sorted) {
    if (!*unsorted) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) {
        const int k = keys[i], id = ids[i];
        perm[i] = id;
        key_sorted[i] = k;
        other_sorted[i] = (int)other[id];
        const int kprev = (i > 0) ? keys[i - 1] : -1;
        for (int r = kprev + 1; r <= k; ++r) rowptr[r] = (int)i;
        if (i == E - 1)
            for (int64_t r = (int64_t)k + 1; r <= Q; ++r) rowptr[r] = (int)E;
    }
}

static int rs_bits(int64_t Q) {
    int b = 1;
    while (((int64_t)1 << b) < Q) ++b;
    return b;
}

template <typename IDX>
int csr_build_t(const IDX* edge_index, int64_t E, int sort_row, int64_t Q, int32_t* rowptr, int32_t* perm,
                int32_t* key_sorted, int32_t* other_sorted, void* ws, hipStream_t st) {
    const IDX* keys = edge_index + (sort_row ? E : 0);
    const IDX* other = edge_index + (sort_row ? 0 : E);
    const int nblk = (int)ceil_div(E, RS_TILE);
    const int64_t tn = (int64_t)256 * nblk;              // [digit][workgroup] table, sized for 8-bit digits
    int* flags = (int*)ws;                  // [4]: flags[1] = unsorted
    int* unsorted = flags + 1;
    int* kbuf0 = flags + 4;                 // [E] x4: key / id ping-pong
    int* kbuf1 = kbuf0 + E;
    int* ibuf0 = kbuf1 + E;
    int* ibuf1 = ibuf0 + E;
    int* table = ibuf1 + E;                 // [tn + 1]
    int* tscan = table + tn + 1;            // [tn + 1]
    int* bsum = tscan + tn + 1;             // [nbt + 1]
    // (a kernel, not hipMemsetAsync: in a replayed hipGraph a memset node costs ~10 us of idle queue on either side of it --
    // profiles/archive/r4_q_graph_replay_gaps.txt)
    GAOT_KLAUNCH(k_zero_flags, dim3(1), dim3(64), 0, st, flags, 4);
    if (E == 0) {
        (void)hipMemsetAsync(rowptr, 0, sizeof(int) * (size_t)(Q + 1), st);
        return GAOT_OK;
    }
    const int tb = 256;
    const int gb = (int)std::min<int64_t>(ceil_div(E, tb), 256 * 16);
    GAOT_KLAUNCH((k_check_sorted<IDX>), dim3(gb), dim3(tb), 0, st, keys, E, unsorted);
    GAOT_KLAUNCH((k_sorted_build<IDX>), dim3(gb), dim3(tb), 0, st, keys, other, E, Q, unsorted, rowptr, perm,
                       key_sorted, other_sorted);
    // general path (every kernel returns at once when the keys were sorted)
    const int bits = rs_bits(Q);
    const int passes = (bits + 7) / 8;
    const int width = (bits + passes - 1) / passes;
    const int* kin = nullptr;
    const int* iin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * width;
        const int nd = 1 << width;
        int* kout = (p & 1) ? kbuf1 : kbuf0;
        int* iout = (p & 1) ? ibuf1 : ibuf0;
        const int64_t n = (int64_t)nd * nblk;
        const int nb = (int)ceil_div(n, SCAN_TILE);
        if (p == 0) GAOT_KLAUNCH((k_rs_hist<IDX>), dim3(nblk), dim3(256), 0, st, keys, E, shift, nd, nblk, table, unsorted);
        else GAOT_KLAUNCH((k_rs_hist<int>), dim3(nblk), dim3(256), 0, st, kin, E, shift, nd, nblk, table, unsorted);
        GAOT_KLAUNCH(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, st, table, n, bsum);
        GAOT_KLAUNCH(k_scan_blocksums, dim3(1), dim3(SCAN_BLOCK), 0, st, bsum, nb);
        GAOT_KLAUNCH(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, st, table, n, bsum, tscan);
        if (p == 0)
            GAOT_KLAUNCH((k_rs_scatter<IDX>), dim3(nblk), dim3(256), 0, st, keys, (const int*)nullptr, E, shift, width,
                               nblk, tscan, kout, iout, unsorted);
        else
            GAOT_KLAUNCH((k_rs_scatter<int>), dim3(nblk), dim3(256), 0, st, kin, iin, E, shift, width, nblk, tscan,
                               kout, iout, unsorted);
        kin = kout;
        iin = iout;
    }
    GAOT_KLAUNCH((k_emit_sorted<IDX>), dim3(gb), dim3(tb), 0, st, kin, iin, other, E, Q, unsorted, rowptr, perm,
                       key_sorted, other_sorted);
    return GAOT_OK;
}

}  // namespace

