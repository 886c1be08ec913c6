// bf16 matrix-core variant of the fused GNO integral transform (precision 1 of gaot_gno_fwd / gaot_gno_bwd).
// Same operator, data flow and determinism as gno.hip (see there for the reference semantics); what changes:
//   * hidden layers and the last layer run on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 8 MFMAs per 64x64
//     layer and 32-edge tile instead of 64 -- the kernel stops being matrix-bound and becomes bound by the
//     erf-GELU VALU work and the row gathers;
//   * layer 0 (the 6 edge coordinates) stays on the exact-fp32 MFMA: coordinates differ in the 3rd decimal between
//     neighbours and must not be rounded to 8 mantissa bits;
//   * the fp32 C/D tile of a layer is rounded to bf16 IN REGISTERS into the two k-step fragments of the next
//     layer (k order 16s + 8(j>>2) + 4*half + (j&3), cdna_hip_programming.md §3); the weights are laid out in LDS
//     (forward) as ready-made per-lane fragments in that same k order, one conflict-free ds_read_b128 each.
#include <stdlib.h>

#include <type_traits>

#include "gno_common.h"

namespace {

using namespace gno;
typedef unsigned short bf16_t;

__device__ __forceinline__ short f2bf(float f) { return __builtin_bit_cast(short, (__bf16)f); }

// fp32 accumulator tile (features on registers) -> the two bf16 operand fragments of the next product
__device__ __forceinline__ void to_frags(const f32x16& v, bf16x8& f0, bf16x8& f1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f0[j] = f2bf(v[j]);
        f1[j] = f2bf(v[8 + j]);
    }
}
// feature index inside a 32-block that element j of lane-half hf of k-step s stands for
__device__ __forceinline__ constexpr int kmap(int s, int j, int hf) { return 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3); }

// =================================================================================================
// Forward
// =================================================================================================
template <int NH>
struct FwdLdsB {
    static constexpr int H = 64, KB = 2;
    static constexpr int w0 = 0;                                  // float [IN0P][H]  (layer 0, fp32, [k][j])
    static constexpr int b0 = w0 + IN0P * H;                      // float [H]
    static constexpr int bl = b0 + H;                             // float (NH-1) x [H]
    static constexpr int bL = bl + (NH - 1) * H;                  // float [32]
    static constexpr int img = bL + 32;                           // bf16 images start here (float index, 16-B aligned)
    static constexpr int img_hidden_floats = KB * KB * 2 * 64 * 8 / 2;   // 8 KB per hidden layer
    static constexpr int imgL = img + (NH - 1) * img_hidden_floats;      // last layer: KB*2 fragments
    static constexpr int imgL_floats = KB * 2 * 64 * 8 / 2;
    static constexpr int weights_end = imgL + imgL_floats;
};

// PIPE: 0 = loads at the point of use, 1 = ids / coordinates pipelined, 2 = and the f rows requested ahead of the MLP.
// NW waves per workgroup share one copy of the weight images: 4 (two workgroups per CU) or 12 (one workgroup: THREE waves per SIMD)
template <int NH, int T, int PIPE = 2, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_gno_fwd_bf16(MlpPtrs mlp, const float* __restrict__ y_pos,
                                                         const float* __restrict__ x_pos, const float* __restrict__ f_y,
                                                         const int* __restrict__ src_s, const int* __restrict__ dst_s,
                                                         const int* __restrict__ rowptr, int64_t E, float* __restrict__ out,
                                                         float* __restrict__ part) {
    constexpr int C = 32, H = 64, KB = 2;
    using L = FwdLdsB<NH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage_all = lds + L::weights_end;             // [NW waves][T][32][C]
    int* ids_all = (int*)(stage_all + NW * T * 32 * C);  // [NW waves][T][2][32]
    constexpr int NTH = NW * 64;

    // ---- layer 0 weights (fp32, transposed) and all biases ------------------------------------------
    for (int i = threadIdx.x; i < IN0P * H; i += NTH) {
        const int k = i / H, j = i % H;
        lds[L::w0 + i] = (k < IN0) ? mlp.w[0][j * IN0 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < H; i += NTH) lds[L::b0 + i] = mlp.b[0][i];
#pragma unroll
    for (int l = 1; l < NH; ++l)
        for (int i = threadIdx.x; i < H; i += NTH) lds[L::bl + (l - 1) * H + i] = mlp.b[l][i];
    for (int i = threadIdx.x; i < C; i += NTH) lds[L::bL + i] = mlp.b[NH][i];
    // ---- bf16 fragment images: hidden layer l, fragment (ob,kb,s), lane, element j ---------------------
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        bf16_t* im = reinterpret_cast<bf16_t*>(lds + L::img + (l - 1) * L::img_hidden_floats);
        for (int i = threadIdx.x; i < KB * KB * 2 * 64 * 8; i += NTH) {
            const int j = i & 7, ln = (i >> 3) & 63, fr = i >> 9;   // fr = (ob*KB + kb)*2 + s
            const int s = fr & 1, kb = (fr >> 1) % KB, ob = (fr >> 1) / KB;
            const int row = 32 * ob + (ln & 31), col = 32 * kb + kmap(s, j, ln >> 5);
            im[i] = (bf16_t)f2bf(mlp.w[l][row * H + col]);
        }
    }
    {
        bf16_t* im = reinterpret_cast<bf16_t*>(lds + L::imgL);
        for (int i = threadIdx.x; i < KB * 2 * 64 * 8; i += NTH) {
            const int j = i & 7, ln = (i >> 3) & 63, fr = i >> 9;   // fr = kb*2 + s
            const int s = fr & 1, kb = fr >> 1;
            im[i] = (bf16_t)f2bf(mlp.w[NH][(ln & 31) * H + 32 * kb + kmap(s, j, ln >> 5)]);
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hf = lane >> 5;
    float* stage = stage_all + wave * (T * 32 * C);
    int* ids = ids_all + wave * (T * 2 * 32);

    const int64_t n_macro = (E + 32 * T - 1) / (32 * T);
    const int64_t mt0 = (int64_t)blockIdx.x * NW + wave, mstep = (int64_t)gridDim.x * NW;
    // Software pipeline over the wave's macro tiles (PIPE): the edge's endpoint ids are loaded TWO tiles ahead and its coordinates
    // gathered ONE tile ahead, so the id -> coordinate -> first MFMA chain (two dependent global round trips at the top of every
    // tile, covered only by the SIMD's other wave) runs under the previous tile's MLP; the f rows of the tile are requested as
    // soon as its ids are in LDS and arrive during the MLP instead of after its last layer.
    int ps[T], pq[T];          // ids of the tile after next (PIPE) / scratch
    float pbin[T][3];          // coordinates of the next tile (PIPE)
    int cs[T], cq[T];          // ids of the next tile (PIPE)
    auto load_ids = [&](int64_t mt, int (&s_)[T], int (&q_)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t e = mt * 32 * T + 32 * t + l31;
            const bool valid = mt < n_macro && e < E;
            s_[t] = valid ? src_s[e] : 0;
            q_[t] = valid ? dst_s[e] : -1;
        }
    };
    auto gather_pos = [&](const int (&s_)[T], const int (&q_)[T], float (&b_)[T][3]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float* ys = y_pos + (int64_t)s_[t] * 3;
            const float* xq = x_pos + (int64_t)(q_[t] < 0 ? 0 : q_[t]) * 3;
            b_[t][0] = ys[hf];
            b_[t][1] = hf ? xq[0] : ys[2];
            b_[t][2] = xq[1 + hf];
        }
    };
    if constexpr (PIPE != 0) {
        load_ids(mt0, cs, cq);
        load_ids(mt0 + mstep, ps, pq);
        gather_pos(cs, cq, pbin);
    }
    for (int64_t mt = mt0; mt < n_macro; mt += mstep) {
        const int64_t base = mt * 32 * T;
        float bin[T][3];
        if constexpr (PIPE != 0) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bin[t][0] = pbin[t][0]; bin[t][1] = pbin[t][1]; bin[t][2] = pbin[t][2];
                if (hf == 0) {
                    ids[(t * 2 + 0) * 32 + l31] = cs[t];
                    ids[(t * 2 + 1) * 32 + l31] = cq[t];
                }
                cs[t] = ps[t]; cq[t] = pq[t];
            }
            gather_pos(cs, cq, pbin);                 // next tile's coordinates (its ids arrived during the previous tile)
            load_ids(mt + 2 * mstep, ps, pq);         // ids of the tile after next
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t e = base + 32 * t + l31;
                const bool valid = e < E;
                const int s = valid ? src_s[e] : 0;
                const int q = valid ? dst_s[e] : 0;
                const float* ys = y_pos + (int64_t)s * 3;
                const float* xq = x_pos + (int64_t)q * 3;
                bin[t][0] = ys[hf];
                bin[t][1] = hf ? xq[0] : ys[2];
                bin[t][2] = xq[1 + hf];
                if (hf == 0) {
                    ids[(t * 2 + 0) * 32 + l31] = s;
                    ids[(t * 2 + 1) * 32 + l31] = valid ? q : -1;
                }
            }
        }
        float fv[PIPE == 2 ? T : 1][16];
        if constexpr (PIPE == 2) {
            wave_lds_fence();  // ids visible to the whole wave
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sidx = ids[(t * 2 + 0) * 32 + mfma32_row(r, hf)];
                    fv[t][r] = f_y[(int64_t)sidx * C + l31];
                }
        }
        // ---- layer 0 on the exact-fp32 MFMA, GELU, round to bf16 fragments ---------------------------
        bf16x8 hb[T][KB][2];
#pragma unroll
        for (int ob = 0; ob < KB; ++ob) {
            f32x16 z[T];
            f32x16 bias;
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = lds[L::b0 + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
            for (int t = 0; t < T; ++t) z[t] = bias;
#pragma unroll
            for (int i = 0; i < IN0 / 2; ++i) {
                const float a = lds[L::w0 + (2 * i + hf) * H + 32 * ob + l31];
#pragma unroll
                for (int t = 0; t < T; ++t) z[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bin[t][i], z[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32v2 y = gelu_e2_2(f32v2{z[t][r], z[t][r + 1]});
                    z[t][r] = y[0]; z[t][r + 1] = y[1];
                }
                to_frags(z[t], hb[t][ob][0], hb[t][ob][1]);
            }
        }
        // ---- hidden layers on the bf16 MFMA --------------------------------------------------------------
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            const bf16x8* im = reinterpret_cast<const bf16x8*>(lds + L::img + (l - 1) * L::img_hidden_floats);
            f32x16 z[T][KB];
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
                f32x16 bias;
#pragma unroll
                for (int r = 0; r < 16; ++r) bias[r] = lds[L::bl + (l - 1) * H + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
                for (int t = 0; t < T; ++t) z[t][ob] = bias;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 a = im[((ob * KB + kb) * 2 + s) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            z[t][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[t][kb][s], z[t][ob], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int ob = 0; ob < KB; ++ob) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32v2 y = gelu_e2_2(f32v2{z[t][ob][r], z[t][ob][r + 1]});
                        z[t][ob][r] = y[0]; z[t][ob][r + 1] = y[1];
                    }
                    to_frags(z[t][ob], hb[t][ob][0], hb[t][ob][1]);
                }
        }
        // ---- last layer, transposed: K'[e][c] = sum_k Hlast[k][e] * WL[c][k] + bL[c] ------------------------
        wave_lds_fence();  // ids visible to the whole wave
        const bf16x8* imL = reinterpret_cast<const bf16x8*>(lds + L::imgL);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x16 acc;
            const float blv = lds[L::bL + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = blv;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb[t][kb][s], imL[(kb * 2 + s) * 64 + lane], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int el = mfma32_row(r, hf);
                if constexpr (PIPE == 2) {
                    stage[(t * 32 + el) * C + l31] = acc[r] * fv[t][r];
                } else {
                    const int s = ids[(t * 2 + 0) * 32 + el];
                    const float fval = f_y[(int64_t)s * C + l31];
                    stage[(t * 32 + el) * C + l31] = acc[r] * fval;
                }
            }
        }
        wave_lds_fence();
        if (hf < T) {
            segment_walk<C>(stage + hf * 32 * C, C, ids + (hf * 2 + 1) * 32, l31, base + 32 * hf, rowptr, out, part, true);
        }
        wave_lds_fence();
    }
}

size_t fwd_lds_bytes_b(int nh, int t, int nw) {
    const int weights = IN0P * 64 + 64 + (nh - 1) * 64 + 32 + (nh - 1) * (2 * 2 * 2 * 64 * 8 / 2) + (2 * 2 * 64 * 8 / 2);
    return sizeof(float) * (size_t)(weights + nw * t * 32 * 32) + sizeof(int) * (size_t)(nw * t * 2 * 32);
}

template <int NH, int PIPE, int NW>
int launch_fwd_v(const MlpPtrs& p, const float* y_pos, const float* x_pos, const float* f_y, const int* src_s,
                 const int* dst_s, const int* rowptr, int64_t E, float* out, float* part, hipStream_t st) {
    constexpr int T = 2;
    const size_t lds = fwd_lds_bytes_b(NH, T, NW);
    auto kern = k_gno_fwd_bf16<NH, T, PIPE, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            gaot_set_error("gno_fwd_bf16: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int64_t n_macro = ceil_div(E, 32 * T);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_macro, NW), NW == 4 ? 256 * 2 : 256));
    GAOT_KLAUNCH(kern, dim3(grid), dim3(NW * 64), lds, st, p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part);
    return GAOT_OK;
}

template <int NH>
int launch_fwd_b(const MlpPtrs& p, const float* y_pos, const float* x_pos, const float* f_y, const int* src_s,
                 const int* dst_s, const int* rowptr, int64_t E, float* out, float* part, hipStream_t st) {
    // Shipped: 12 waves per workgroup (one weight image per CU, THREE waves per SIMD; 127 KB of LDS at three hidden layers) with the
    // id / coordinate loads pipelined.  Measured at E = 4 M (profiles/archive/r5_w_gno_fwd_variants_lab.txt; nh = 3 / 2 / 4): round-4 form
    // 0.402 / 0.317 / 0.466 ms, 4 waves + full pipeline 0.390 / 0.300 / 0.435, 12 waves + id pipeline 0.377 / 0.277 / 0.421,
    // 12 waves alone 0.377 / 0.279 / 0.427.  GAOT_GNO_FWD_VARIANT (measurement only) = 0: 4 waves, no pipeline | 1: 4 waves, ids +
    // coordinates + f rows pipelined | 2: the shipped form | 3: 12 waves, no pipeline
    static const int variant = [] { const char* e = getenv("GAOT_GNO_FWD_VARIANT"); return e ? atoi(e) : 2; }();
    switch (variant) {
        case 0: return launch_fwd_v<NH, 0, 4>(p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part, st);
        case 1: return launch_fwd_v<NH, 2, 4>(p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part, st);
        case 3: return launch_fwd_v<NH, 0, 12>(p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part, st);
        default: return launch_fwd_v<NH, 1, 12>(p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part, st);
    }
}

}  // namespace

// called from gaot_gno_fwd (gno.hip) when precision == 1; num_edges > 0, shapes already validated
int gaot_gno_fwd_bf16_dispatch(int n_hidden, const float* const* w, const float* const* b, const float* y_pos,
                               const float* x_pos, const float* f_y, const int32_t* src_sorted, const int32_t* dst_sorted,
                               const int32_t* rowptr_dst, int64_t num_edges, float* out, float* part, hipStream_t st) {
    MlpPtrs p;
    for (int l = 0; l <= n_hidden; ++l) { p.w[l] = w[l]; p.b[l] = b[l]; }
    switch (n_hidden) {
        case 1: return launch_fwd_b<1>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st);
        case 2: return launch_fwd_b<2>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st);
        case 3: return launch_fwd_b<3>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st);
        case 4: return launch_fwd_b<4>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st);
    }
    gaot_set_error("gaot_gno_fwd (bf16): unsupported n_hidden %d", n_hidden);
    return GAOT_ERR_UNSUPPORTED;
}

// =================================================================================================
// Backward (bf16 matrix cores): k_gno_bwd3_bf16 (gno_bwd3_bf16.hip)
// =================================================================================================
int gaot_gno_bwd3_bf16_launch(int n_hidden, void* images, const float* w0t, const float* const* w, const float* const* b,
                              const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                              const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                              int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st);

// recompute and transposed data-gradient fragment images of the hidden layers (8 KB each) and of the last layer (4 KB each)
size_t gaot_gno_bwd_bf16_image_bytes(int n_hidden) { return (size_t)(n_hidden - 1) * 2 * 8192 + 2 * 4096 + 256; }

// called from gaot_gno_bwd (gno.hip) for precision == 1.  `images` = scratch of gaot_gno_bwd_bf16_image_bytes();
// w0t = fp32 [6][64] transposed first-layer weight (prepared by the caller, shared with the fp32 path).
int gaot_gno_bwd_bf16_dispatch(int n_hidden, const float* const* w, const float* const* b, const float* w0t,
                               void* images, const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                               const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                               int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    return gaot_gno_bwd3_bf16_launch(n_hidden, images, w0t, w, b, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src,
                                     num_edges, grad_f, part, wpart, grid, st);
}
