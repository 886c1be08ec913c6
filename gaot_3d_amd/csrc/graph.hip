// Neighbour-graph construction on the device: the reference builds its bipartite phys<->latent graphs
// with torch_cluster knn / radius on the CPU (src/model/layers/magno.py:116-295, callers collate_functions.py:73-130,
// stat.py:176-214).  The latent tokens are a REGULAR D x H x W grid (gaot_3d.py:35-46, stat.py:238-252), so both
// searches are closed-form cell lookups around the point's nearest grid node -- O(N * window), no tree, no sort:
//   knn    : pyg_knn(x=latent, y=phys, k)        -> for every point its k nearest tokens        (magno.py:183-189, 241-249)
//   radius : pyg_radius(x=latent, y=phys, r)     -> for every point the tokens within r, <= cap (magno.py:253-261)
//            pyg_radius(x=phys, y=latent, r)     -> same pairs grouped by token, <= cap points per token (193-201):
//                                                   enumerate per point, stable-sort by token (csr_impl.h), keep the
//                                                   first cap points of every token, compact
//   'bidirectional' = coalesce(cat(knn, radius)) (219-220, 292-293): two stable sorts + adjacent-unique + compact.
// Distances use the ACTUAL token coordinates (gathered from the token array), the grid descriptor only bounds the
// search window; the window grows until the k-th distance is provably inside it, so the result is exact.
// Tie rules (torch_cluster leaves them to the implementation): knn orders by (distance, token index); radius keeps the
// `cap` lowest indices, `d <= r`.  All integer / HBM-bound work; every list is produced in a deterministic order.
#include <cfloat>

#include "csr_impl.h"

namespace {

struct GridDesc {
    int dim[3];
    float lo[3], step[3], inv_step[3];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float dist2(const float* __restrict__ tok, int lin, float px, float py, float pz) {
    const float dx = tok[3 * lin] - px, dy = tok[3 * lin + 1] - py, dz = tok[3 * lin + 2] - pz;
    return dx * dx + dy * dy + dz * dz;
}

template <int K>
__global__ __launch_bounds__(256) void k_knn_grid(const float* __restrict__ pos, int64_t N, GridDesc g,
                                                  const float* __restrict__ tok, int* __restrict__ out, int kout) {
    // K = capacity of the register list (an instantiated size >= kout); the first kout entries of the sorted top-K list ARE
    // the top-kout list, so any k <= 64 runs on the next instantiated capacity (torch_cluster takes any k, magno.py:183-189)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    float f[3];
    int base[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f[a] = (p[a] - g.lo[a]) * g.inv_step[a];
        base[a] = clampi((int)rintf(f[a]), 0, g.dim[a] - 1);
    }
    float bd[K];
    int bi[K];
    for (int w = 1;; ++w) {
#pragma unroll
        for (int j = 0; j < K; ++j) { bd[j] = FLT_MAX; bi[j] = 0x7fffffff; }
        int c0[3], c1[3];
        bool whole = true;
        float bound = FLT_MAX;   // no token outside the window is closer than this
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c0[a] = base[a] - w > 0 ? base[a] - w : 0;
            c1[a] = base[a] + w < g.dim[a] - 1 ? base[a] + w : g.dim[a] - 1;
            if (c0[a] > 0) { whole = false; bound = fminf(bound, (f[a] - (float)(c0[a] - 1)) * g.step[a]); }
            if (c1[a] < g.dim[a] - 1) { whole = false; bound = fminf(bound, ((float)(c1[a] + 1) - f[a]) * g.step[a]); }
        }
        for (int cd = c0[0]; cd <= c1[0]; ++cd)
            for (int ch = c0[1]; ch <= c1[1]; ++ch)
                for (int cw = c0[2]; cw <= c1[2]; ++cw) {
                    const int lin = (cd * g.dim[1] + ch) * g.dim[2] + cw;
                    const float d = dist2(tok, lin, p[0], p[1], p[2]);
                    if (d < bd[K - 1] || (d == bd[K - 1] && lin < bi[K - 1])) {
                        bd[K - 1] = d;
                        bi[K - 1] = lin;
#pragma unroll
                        for (int j = K - 1; j > 0; --j) {
                            const bool lt = bd[j] < bd[j - 1] || (bd[j] == bd[j - 1] && bi[j] < bi[j - 1]);
                            if (lt) {
                                const float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
                                const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
                            }
                        }
                    }
                }
        // exact once the k-th distance is strictly inside the window (1e-4 relative slack covers the rounding of the
        // grid coordinates against the stored token coordinates), or the window is the whole grid
        const float b = bound * (1.0f - 1e-4f);
        float kth = bd[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) kth = (j == kout - 1) ? bd[j] : kth;
        if (whole || (bound > 0.f && kth < b * b)) break;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < kout) out[i * kout + j] = bi[j];
}

// tokens within `radius` of point i, ascending token index; FILL == false: counts[i] = min(count, cap);
// FILL == true: writes (i, token) pairs at offsets[i]
template <bool FILL>
__global__ __launch_bounds__(256) void k_radius_grid(const float* __restrict__ pos, int64_t N, GridDesc g,
                                                     const float* __restrict__ tok, float radius, int cap,
                                                     int* __restrict__ counts, const int* __restrict__ offsets,
                                                     int* __restrict__ out_center, int* __restrict__ out_other) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    const float p[3] = {px, py, pz};
    int c0[3], c1[3];
    const float rr = radius * (1.0f + 1e-4f);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float f = (p[a] - g.lo[a]) * g.inv_step[a];
        const float wr = rr * g.inv_step[a];
        c0[a] = clampi((int)floorf(f - wr), 0, g.dim[a] - 1);
        c1[a] = clampi((int)ceilf(f + wr), 0, g.dim[a] - 1);
        if (g.dim[a] == 1) { c0[a] = 0; c1[a] = 0; }
    }
    const float r2 = radius * radius;
    int n = 0;
    const int o = FILL ? offsets[i] : 0;
    for (int cd = c0[0]; cd <= c1[0] && n < cap; ++cd)
        for (int ch = c0[1]; ch <= c1[1] && n < cap; ++ch)
            for (int cw = c0[2]; cw <= c1[2] && n < cap; ++cw) {
                const int lin = (cd * g.dim[1] + ch) * g.dim[2] + cw;
                if (dist2(tok, lin, px, py, pz) <= r2) {
                    if (FILL) { out_center[o + n] = (int)i; out_other[o + n] = lin; }
                    ++n;
                }
            }
    if (!FILL) counts[i] = n;
}

// ---- token sets that are NOT a regular grid (custom `tokens_pos`: the reference's get_neighbor_strategy takes any latent
// coordinates, magno.py:116-124): every point scans ALL tokens, staged through LDS in tiles -- O(N M) distance evaluations on
// the device (6.5 x 10^10 at configs[1] sizes: milliseconds) instead of an N x M distance matrix in host or device memory.
// Same distance arithmetic, order and tie rules as the grid kernels above, so the lists are identical where both apply.
constexpr int BRUTE_TILE = 2048;

template <int K>
__global__ __launch_bounds__(256) void k_knn_brute(const float* __restrict__ pos, int64_t N, const float* __restrict__ tok,
                                                   int64_t M, int* __restrict__ out, int kout) {
    __shared__ float ts[BRUTE_TILE * 3];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    const float px = live ? pos[3 * i] : 0.f, py = live ? pos[3 * i + 1] : 0.f, pz = live ? pos[3 * i + 2] : 0.f;
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = FLT_MAX; bi[j] = 0x7fffffff; }
    for (int64_t t0 = 0; t0 < M; t0 += BRUTE_TILE) {
        const int nt = (int)min((int64_t)BRUTE_TILE, M - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < nt * 3; j += 256) ts[j] = tok[t0 * 3 + j];
        __syncthreads();
        for (int j = 0; j < nt; ++j) {
            const float d = dist2(ts, j, px, py, pz);
            const int lin = (int)(t0 + j);
            if (d < bd[K - 1] || (d == bd[K - 1] && lin < bi[K - 1])) {
                bd[K - 1] = d;
                bi[K - 1] = lin;
#pragma unroll
                for (int q = K - 1; q > 0; --q) {
                    const bool lt = bd[q] < bd[q - 1] || (bd[q] == bd[q - 1] && bi[q] < bi[q - 1]);
                    if (lt) {
                        const float td = bd[q]; bd[q] = bd[q - 1]; bd[q - 1] = td;
                        const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
                    }
                }
            }
        }
    }
    if (live)
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < kout) out[i * kout + j] = bi[j];      // K = capacity >= kout (see k_knn_grid)
}

// k > 64 (torch_cluster takes any k, magno.py:183-189): selection in passes of 64.  Pass `start` picks the 64 nearest tokens whose
// key (distance, index) lies strictly BEHIND the key of entry start-1 -- the last one the previous pass wrote -- so the passes
// together produce the same (distance, index)-ordered list one big register list would.  The distance is formed without
// contraction (__fmul_rn / __fadd_rn), so the floor's key recomputed here equals the key it was selected with, whatever the
// compiler does around it.
__device__ __forceinline__ float dist2_rn(const float* __restrict__ tok, int64_t lin, float px, float py, float pz) {
    const float dx = __fsub_rn(tok[3 * lin], px), dy = __fsub_rn(tok[3 * lin + 1], py), dz = __fsub_rn(tok[3 * lin + 2], pz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__global__ __launch_bounds__(256) void k_knn_brute_pass(const float* __restrict__ pos, int64_t N, const float* __restrict__ tok,
                                                        int64_t M, int* __restrict__ out, int kout, int start) {
    constexpr int K = 64;
    __shared__ float ts[BRUTE_TILE * 3];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    const float px = live ? pos[3 * i] : 0.f, py = live ? pos[3 * i + 1] : 0.f, pz = live ? pos[3 * i + 2] : 0.f;
    float fd = -1.f;          // floor key: nothing selected yet
    int fi = -1;
    if (live && start > 0) {
        fi = out[i * kout + start - 1];
        fd = dist2_rn(tok, fi, px, py, pz);
    }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = FLT_MAX; bi[j] = 0x7fffffff; }
    for (int64_t t0 = 0; t0 < M; t0 += BRUTE_TILE) {
        const int nt = (int)min((int64_t)BRUTE_TILE, M - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < nt * 3; j += 256) ts[j] = tok[t0 * 3 + j];
        __syncthreads();
        for (int j = 0; j < nt; ++j) {
            const float d = dist2_rn(ts, j, px, py, pz);
            const int lin = (int)(t0 + j);
            const bool behind = d > fd || (d == fd && lin > fi);
            if (behind && (d < bd[K - 1] || (d == bd[K - 1] && lin < bi[K - 1]))) {
                bd[K - 1] = d;
                bi[K - 1] = lin;
#pragma unroll
                for (int q = K - 1; q > 0; --q) {
                    const bool lt = bd[q] < bd[q - 1] || (bd[q] == bd[q - 1] && bi[q] < bi[q - 1]);
                    if (lt) {
                        const float td = bd[q]; bd[q] = bd[q - 1]; bd[q - 1] = td;
                        const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
                    }
                }
            }
        }
    }
    if (live)
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (start + j < kout) out[i * kout + start + j] = bi[j];
}
static void knn_many(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens, int k, int32_t* out_idx,
                     hipStream_t st) {
    const dim3 grd((unsigned)ceil_div(num_points, 256)), blk(256);
    for (int start = 0; start < k; start += 64)
        GAOT_KLAUNCH(k_knn_brute_pass, grd, blk, 0, st, pos, num_points, token_pos, num_tokens, out_idx, k, start);
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_radius_brute(const float* __restrict__ pos, int64_t N, const float* __restrict__ tok,
                                                      int64_t M, float radius, int cap, int* __restrict__ counts,
                                                      const int* __restrict__ offsets, int* __restrict__ out_center,
                                                      int* __restrict__ out_other) {
    __shared__ float ts[BRUTE_TILE * 3];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    const float px = live ? pos[3 * i] : 0.f, py = live ? pos[3 * i + 1] : 0.f, pz = live ? pos[3 * i + 2] : 0.f;
    const float r2 = radius * radius;
    int n = 0;
    const int o = (FILL && live) ? offsets[i] : 0;
    for (int64_t t0 = 0; t0 < M; t0 += BRUTE_TILE) {
        const int nt = (int)min((int64_t)BRUTE_TILE, M - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < nt * 3; j += 256) ts[j] = tok[t0 * 3 + j];
        __syncthreads();
        if (live)
            for (int j = 0; j < nt && n < cap; ++j)
                if (dist2(ts, j, px, py, pz) <= r2) {
                    if (FILL) { out_center[o + n] = (int)i; out_other[o + n] = (int)(t0 + j); }
                    ++n;
                }
    }
    if (!FILL && live) counts[i] = n;
}

// flags[i] = 1 when element i of a key-sorted list is among the first `cap` of its segment
__global__ void k_segment_cap_flags(const int* __restrict__ rowptr, const int* __restrict__ key_sorted, int64_t P, int cap,
                                    int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    flags[i] = (i - rowptr[key_sorted[i]] < cap) ? 1 : 0;
}

// ---- neighbour sampling (reference magno.py:297-371, apply_neighbor_sampling) -------------------------------
// The reference draws from torch's generator (randperm per over-full query; torch_geometric dropout_edge); here the
// draw is a counter-based 32-bit hash of (seed, edge id), restated in the oracle, so a sample is reproducible from
// its seed word and needs no generator state on the device.
__device__ __forceinline__ uint32_t sample_hash(unsigned long long seed, uint32_t i) {
    uint32_t x = (uint32_t)seed ^ (i * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    x += (uint32_t)(seed >> 32);
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15;
    return x;
}
// element dropout (nn.Dropout of the reference's channel MLPs, mlp.py:268-272, 318-322): out = keep ? x / (1 - p) : 0 with
// keep(i) = sample_hash(seed, i) >= thr; the backward applies the same mask to the gradient
__global__ void k_dropout(const float* __restrict__ x, const unsigned long long* __restrict__ seed, uint32_t thr, float scale,
                          int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sample_hash(*seed, (uint32_t)i) >= thr ? x[i] * scale : 0.f;
}
// 'ratio': keep edge i with probability thr / 2^32
__global__ void k_random_keep_flags(const unsigned long long* __restrict__ seed, int64_t n, uint32_t thr, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = sample_hash(*seed, (uint32_t)i) < thr ? 1 : 0;
}
// 'max_neighbors': of every segment with more than cap entries keep the cap entries with the smallest (hash, index)
// -- a uniformly random subset; smaller segments are kept whole
__global__ void k_segment_random_cap_flags(const unsigned long long* __restrict__ seed, const int* __restrict__ rowptr,
                                           const int* __restrict__ key_sorted, int64_t n, int cap, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = key_sorted[i];
    const int lo = rowptr[r], hi = rowptr[r + 1];
    if (hi - lo <= cap) { flags[i] = 1; return; }
    const unsigned long long sd = *seed;
    const uint32_t mine = sample_hash(sd, (uint32_t)i);
    int rank = 0;
    for (int j = lo; j < hi; ++j) {
        const uint32_t h = sample_hash(sd, (uint32_t)j);
        rank += (h < mine || (h == mine && j < i)) ? 1 : 0;
    }
    flags[i] = rank < cap ? 1 : 0;
}

// flags[i] = 1 when pair i differs from pair i-1 (lists sorted by (a, b))
__global__ void k_unique_pair_flags(const int* __restrict__ a, const int* __restrict__ b, int64_t P, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    flags[i] = (i == 0 || a[i] != a[i - 1] || b[i] != b[i - 1]) ? 1 : 0;
}

__global__ void k_compact_pairs(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ flags,
                                const int* __restrict__ offsets, int64_t P, int* __restrict__ out_a, int* __restrict__ out_b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P || !flags[i]) return;
    out_a[offsets[i]] = a[i];
    out_b[offsets[i]] = b[i];
}

int make_grid(const gaot_grid_t* grid, GridDesc& g) {
    if (!grid) return 1;
    for (int a = 0; a < 3; ++a) {
        if (grid->dims[a] < 1) return 1;
        g.dim[a] = grid->dims[a];
        g.lo[a] = grid->lo[a];
        const float span = grid->hi[a] - grid->lo[a];
        if (grid->dims[a] > 1 && !(span > 0.f)) return 1;
        g.step[a] = grid->dims[a] > 1 ? span / (float)(grid->dims[a] - 1) : 1.0f;
        g.inv_step[a] = 1.0f / g.step[a];
    }
    return 0;
}

}  // namespace

extern "C" int gaot_knn_grid(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos, int k,
                             int32_t* out_idx, gaot_stream_t stream) {
    GAOT_ENTER();
    GridDesc g;
    GAOT_CHECK_ARG(make_grid(grid, g) == 0, "bad grid descriptor");
    GAOT_CHECK_ARG(num_points >= 0, "negative size");
    const int64_t m = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    GAOT_CHECK_ARG(k >= 1 && k <= m, "k must be in [1, number of tokens]");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && token_pos && out_idx, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (k > 64) {     // beyond the register list: passes of 64 over all tokens (k_knn_brute_pass)
        knn_many(pos, num_points, token_pos, m, k, out_idx, st);
        GAOT_LAUNCH_CHECK();
        return GAOT_OK;
    }
    const dim3 grd((unsigned)ceil_div(num_points, 256)), blk(256);
#define GAOT_KNN(KK) GAOT_KLAUNCH((k_knn_grid<KK>), grd, blk, 0, st, pos, num_points, g, token_pos, out_idx, k)
    switch (k <= 8 ? k : k <= 12 ? 12 : k <= 16 ? 16 : k <= 24 ? 24 : k <= 32 ? 32 : k <= 48 ? 48 : 64) {
        case 1: GAOT_KNN(1); break;
        case 2: GAOT_KNN(2); break;
        case 3: GAOT_KNN(3); break;
        case 4: GAOT_KNN(4); break;
        case 5: GAOT_KNN(5); break;
        case 6: GAOT_KNN(6); break;
        case 7: GAOT_KNN(7); break;
        case 8: GAOT_KNN(8); break;
        case 12: GAOT_KNN(12); break;
        case 16: GAOT_KNN(16); break;
        case 24: GAOT_KNN(24); break;
        case 32: GAOT_KNN(32); break;
        case 48: GAOT_KNN(48); break;
        default: GAOT_KNN(64); break;
    }
#undef GAOT_KNN
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_knn_brute(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens, int k,
                              int32_t* out_idx, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_points >= 0 && num_tokens >= 1, "bad size");
    GAOT_CHECK_ARG(k >= 1 && k <= num_tokens, "k must be in [1, number of tokens]");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && token_pos && out_idx, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (k > 64) {
        knn_many(pos, num_points, token_pos, num_tokens, k, out_idx, st);
        GAOT_LAUNCH_CHECK();
        return GAOT_OK;
    }
    const dim3 grd((unsigned)ceil_div(num_points, 256)), blk(256);
#define GAOT_KNNB(KK) GAOT_KLAUNCH((k_knn_brute<KK>), grd, blk, 0, st, pos, num_points, token_pos, num_tokens, out_idx, k)
    switch (k <= 8 ? k : k <= 12 ? 12 : k <= 16 ? 16 : k <= 24 ? 24 : k <= 32 ? 32 : k <= 48 ? 48 : 64) {
        case 1: GAOT_KNNB(1); break;
        case 2: GAOT_KNNB(2); break;
        case 3: GAOT_KNNB(3); break;
        case 4: GAOT_KNNB(4); break;
        case 5: GAOT_KNNB(5); break;
        case 6: GAOT_KNNB(6); break;
        case 7: GAOT_KNNB(7); break;
        case 8: GAOT_KNNB(8); break;
        case 12: GAOT_KNNB(12); break;
        case 16: GAOT_KNNB(16); break;
        case 24: GAOT_KNNB(24); break;
        case 32: GAOT_KNNB(32); break;
        case 48: GAOT_KNNB(48); break;
        default: GAOT_KNNB(64); break;
    }
#undef GAOT_KNNB
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_radius_brute_count(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens,
                                       float radius, int cap, int32_t* counts, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_points >= 0 && num_tokens >= 0 && radius >= 0.f && cap >= 1, "bad argument");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && counts && (token_pos || num_tokens == 0), "null pointer");
    GAOT_KLAUNCH((k_radius_brute<false>), dim3((unsigned)ceil_div(num_points, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                 num_points, token_pos, num_tokens, radius, cap, counts, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_radius_brute_fill(const float* pos, int64_t num_points, const float* token_pos, int64_t num_tokens,
                                      float radius, int cap, const int32_t* offsets, int32_t* out_point, int32_t* out_token,
                                      gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_points >= 0 && num_tokens >= 0 && radius >= 0.f && cap >= 1, "bad argument");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && offsets && out_point && out_token, "null pointer");
    GAOT_KLAUNCH((k_radius_brute<true>), dim3((unsigned)ceil_div(num_points, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                 num_points, token_pos, num_tokens, radius, cap, (int*)nullptr, offsets, out_point, out_token);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_radius_grid_count(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos,
                                      float radius, int cap, int32_t* counts, gaot_stream_t stream) {
    GAOT_ENTER();
    GridDesc g;
    GAOT_CHECK_ARG(make_grid(grid, g) == 0, "bad grid descriptor");
    GAOT_CHECK_ARG(num_points >= 0 && radius >= 0.f && cap >= 1, "bad argument");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && token_pos && counts, "null pointer");
    GAOT_KLAUNCH((k_radius_grid<false>), dim3((unsigned)ceil_div(num_points, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       num_points, g, token_pos, radius, cap, counts, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_radius_grid_fill(const float* pos, int64_t num_points, const gaot_grid_t* grid, const float* token_pos,
                                     float radius, int cap, const int32_t* offsets, int32_t* out_point, int32_t* out_token,
                                     gaot_stream_t stream) {
    GAOT_ENTER();
    GridDesc g;
    GAOT_CHECK_ARG(make_grid(grid, g) == 0, "bad grid descriptor");
    GAOT_CHECK_ARG(num_points >= 0 && radius >= 0.f && cap >= 1, "bad argument");
    if (num_points == 0) return GAOT_OK;
    GAOT_CHECK_ARG(pos && token_pos && offsets && out_point && out_token, "null pointer");
    GAOT_KLAUNCH((k_radius_grid<true>), dim3((unsigned)ceil_div(num_points, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       num_points, g, token_pos, radius, cap, (int*)nullptr, offsets, out_point, out_token);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" size_t gaot_exclusive_scan_workspace_bytes(int64_t n) { return sizeof(int) * (size_t)(ceil_div(n, SCAN_TILE) + 2) + 64; }

extern "C" int gaot_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* workspace, size_t workspace_bytes,
                                       gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0, "negative size");
    GAOT_CHECK_ARG(out, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        (void)hipMemsetAsync(out, 0, sizeof(int), st);
        return GAOT_OK;
    }
    GAOT_CHECK_ARG(in && workspace && workspace_bytes >= gaot_exclusive_scan_workspace_bytes(n), "workspace too small");
    const int nb = (int)ceil_div(n, SCAN_TILE);
    int* bsum = (int*)workspace;
    GAOT_KLAUNCH(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, st, in, n, bsum);
    GAOT_KLAUNCH(k_scan_blocksums, dim3(1), dim3(SCAN_BLOCK), 0, st, bsum, nb);
    GAOT_KLAUNCH(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, st, in, n, bsum, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_cap_flags(const int32_t* rowptr, const int32_t* key_sorted, int64_t n, int cap, int32_t* flags,
                                      gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && cap >= 1, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(rowptr && key_sorted && flags, "null pointer");
    GAOT_KLAUNCH(k_segment_cap_flags, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, key_sorted,
                       n, cap, flags);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_random_keep_flags(const unsigned long long* seed, int64_t n, double keep_prob, int32_t* flags,
                                      gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && keep_prob >= 0.0 && keep_prob <= 1.0, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(seed && flags, "null pointer");
    double t = keep_prob * 4294967296.0 + 0.5;
    const uint32_t thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    GAOT_KLAUNCH(k_random_keep_flags, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, n, thr, flags);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_dropout(const float* x, const uint64_t* seed_, double p, int64_t n, float* out, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && p >= 0.0 && p < 1.0, "bad argument");
    GAOT_CHECK_ARG(n < ((int64_t)1 << 32), "more than 2^32 elements");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(x && seed_ && out, "null pointer");
    const unsigned long long* seed = (const unsigned long long*)seed_;
    double t = p * 4294967296.0 + 0.5;
    const uint32_t thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    GAOT_KLAUNCH(k_dropout, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, seed, thr,
                 (float)(1.0 / (1.0 - p)), n, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_random_cap_flags(const unsigned long long* seed, const int32_t* rowptr, const int32_t* key_sorted,
                                             int64_t n, int cap, int32_t* flags, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && cap >= 1, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(seed && rowptr && key_sorted && flags, "null pointer");
    GAOT_KLAUNCH(k_segment_random_cap_flags, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       rowptr, key_sorted, n, cap, flags);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_unique_pair_flags(const int32_t* a, const int32_t* b, int64_t n, int32_t* flags, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0, "negative size");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(a && b && flags, "null pointer");
    GAOT_KLAUNCH(k_unique_pair_flags, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, flags);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_compact_pairs(const int32_t* a, const int32_t* b, const int32_t* flags, const int32_t* offsets, int64_t n,
                                  int32_t* out_a, int32_t* out_b, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0, "negative size");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(a && b && flags && offsets && out_a && out_b, "null pointer");
    GAOT_KLAUNCH(k_compact_pairs, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, flags, offsets, n,
                       out_a, out_b);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
