// Pieces shared by the fp32 (gno.hip) and bf16 (gno_bf16.hip) GNO kernels: the per-lane segmented walk over a
// 32-edge LDS tile, the tile-ordered fix-up of rows that straddle tiles, small structs.
#pragma once
#include "common.h"

namespace gno {

constexpr int IN0 = 6;      // [y_pos(3), x_pos(3)]
constexpr int IN0P = 8;     // padded

__device__ __forceinline__ void wave_lds_fence() {
    // LDS ops of one wave execute in order; this only stops the compiler from reordering.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct MlpPtrs {
    const float* w[GAOT_MAX_MLP_LAYERS];
    const float* b[GAOT_MAX_MLP_LAYERS];
};
struct MlpGradPtrs {
    float* w[GAOT_MAX_MLP_LAYERS];
    float* b[GAOT_MAX_MLP_LAYERS];
};

// Per-lane walk over one 32-edge tile staged in LDS as stage[e*ld + c]: lane = channel c.
// Complete rows are stored (mean or sum); rows open to the left go to part slot 0, rows open only
// to the right to slot 1 (combined later by k_segment_fixup in tile order).
template <int C>
__device__ __forceinline__ void segment_walk(const float* stage, int ld, const int* ids, int c, int64_t tbase,
                                             const int* __restrict__ rowptr, float* __restrict__ out,
                                             float* __restrict__ part, bool mean) {
    const int64_t tile = tbase >> 5;
    int cur = ids[0];
    float sum = 0.f;
    auto flush = [&](int q, float s) {
        if (q < 0) return;
        const int rb = rowptr[q], re = rowptr[q + 1];
        const bool ol = rb < tbase, orr = re > tbase + 32;
        if (!ol && !orr) {
            out[(int64_t)q * C + c] = mean ? s / (float)(re - rb) : s;
        } else if (ol) {
            part[(tile * 2 + 0) * C + c] = s;
        } else {
            part[(tile * 2 + 1) * C + c] = s;
        }
    };
#pragma unroll 4
    for (int e = 0; e < 32; ++e) {
        const int d = ids[e];
        if (d != cur) {
            flush(cur, sum);
            cur = d;
            sum = 0.f;
        }
        if (d >= 0) sum += stage[e * ld + c];
    }
    flush(cur, sum);
}

// rows with no edges -> 0; rows spanning several tiles -> ordered sum of the tile partials (tiles of 2^SHIFT edges: 32
// for the forward and the fp32 backward, 16 for the bf16 backward, whose waves own 16-edge tiles)
template <int C, int SHIFT = 5>
__global__ void k_segment_fixup(const int* __restrict__ rowptr, int64_t Q, const float* __restrict__ part,
                                float* __restrict__ out, int mean) {
    static_assert(C % 4 == 0, "four channels per thread");
    constexpr int TPR = C / 4;                                   // threads per row, 16 bytes each
    // grid-stride over (row, 16-byte column group): nearly every row lies inside one tile and needs nothing, so the pass is a scan
    // of rowptr -- launched as one thread per item it was bound by the rate at which workgroups start (8 M point rows: 250 K
    // workgroups, 333 / 623 us), not by the 32 MB it reads
    const int64_t n = Q * TPR, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t q = i / TPR;
        const int c = (int)(i % TPR) * 4;
        const int rb = rowptr[q], re = rowptr[q + 1];
        float4* o = reinterpret_cast<float4*>(out + q * C + c);
        if (re == rb) {
            *o = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int t0 = rb >> SHIFT, t1 = (re - 1) >> SHIFT;
        if (t0 == t1) continue;
        float4 s = *reinterpret_cast<const float4*>(part + ((int64_t)t0 * 2 + 1) * C + c);
        for (int t = t0 + 1; t <= t1; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)t * 2 + 0) * C + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (mean) {
            const float d = (float)(re - rb);
            s.x /= d; s.y /= d; s.z /= d; s.w /= d;
        }
        *o = s;
    }
}
// grid of the fix-up pass: one thread per item up to 16 workgroups per CU, grid-stride beyond
inline unsigned segment_fixup_grid(int64_t items) {
    const int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace gno
