// Fused GNO kernel integral transform for gfx950 -- forward and backward.
//
// Reference semantics (src/model/layers/integral_transform.py:114-171, transform_type="linear",
// reduction="mean"; kernel MLP = LinearChannelMLP, src/model/layers/mlp.py:327-335):
//     k_e   = MLP([y_pos[src_e], x_pos[dst_e]])          6 -> H (xNH, erf-GELU) -> C
//     out_q = (1/deg_q) * sum_{e: dst_e = q} k_e * f_y[src_e]          (deg_q = 0 -> 0)
// The reference materialises [E,6], NH x [E,H] and 3 x [E,C] tensors in HBM; here one launch
// gathers, runs the MLP on the matrix cores and reduces, so HBM sees 32 + C*4 bytes per edge.
//
// Mapping to CDNA4 (wave = 64 lanes, v_mfma_f32_32x32x2_f32 = exact fp32, 64 FLOP/clk/SIMD):
//   * a tile = 32 consecutive edges of the row-sorted list; EDGES sit on the MFMA column (lane)
//     axis, FEATURES on the row (register) axis:  Z[j][e] = sum_k W[j][k] * Hprev[k][e].
//     The C/D tile of one layer (16 regs/lane) is directly the B operand of the next layer (the
//     k order is permuted identically on the weight side), so activations never leave registers.
//   * the last layer is computed transposed (A = activations, B = W^T) so that CHANNELS land on
//     lanes: the f_y[src] row gather is then one coalesced 128-B read per half-wave and the
//     segmented sum over edges is a per-lane walk over an LDS tile.
//   * weights sit transposed in LDS ([in][out]) -> every A-operand fetch is a conflict-free
//     ds_read_b32 across lanes.
//   * rows that straddle tiles are combined through a small partial buffer in tile order
//     (deterministic; no float atomics anywhere).
#include <type_traits>

#include "gno_common.h"

namespace {

using namespace gno;

// =================================================================================================
// Forward
// =================================================================================================
template <int NH, int H>
struct FwdLds {
    static constexpr int KB = H / 32;
    static constexpr int w0 = 0;                       // [IN0P][H]
    static constexpr int b0 = w0 + IN0P * H;           // [H]
    static constexpr int wl = b0 + H;                  // (NH-1) x ([H][H] + [H])
    static constexpr int wL = wl + (NH - 1) * (H * H + H);  // [H][32]
    static constexpr int bL = wL + H * 32;             // [32]
    static constexpr int weights_end = bL + 32;
};

template <int NH, int H, int T>
__global__ __launch_bounds__(256, 2) void k_gno_fwd(MlpPtrs mlp, const float* __restrict__ y_pos,
                                                    const float* __restrict__ x_pos, const float* __restrict__ f_y,
                                                    const int* __restrict__ src_s, const int* __restrict__ dst_s,
                                                    const int* __restrict__ rowptr, int64_t E, float* __restrict__ out,
                                                    float* __restrict__ part) {
    constexpr int C = 32;
    constexpr int KB = H / 32;
    using L = FwdLds<NH, H>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage_all = lds + L::weights_end;             // [4 waves][T][32][C]
    int* ids_all = (int*)(stage_all + 4 * T * 32 * C);   // [4 waves][T][2][32]

    // ---- weights -> LDS, transposed to [in][out] ------------------------------------------------
    for (int i = threadIdx.x; i < IN0P * H; i += 256) {
        const int k = i / H, j = i % H;
        lds[L::w0 + i] = (k < IN0) ? mlp.w[0][j * IN0 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < H; i += 256) lds[L::b0 + i] = mlp.b[0][i];
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        float* wt = lds + L::wl + (l - 1) * (H * H + H);
        for (int i = threadIdx.x; i < H * H; i += 256) {
            const int j = i / H, k = i % H;
            wt[k * H + j] = mlp.w[l][i];
        }
        for (int i = threadIdx.x; i < H; i += 256) wt[H * H + i] = mlp.b[l][i];
    }
    for (int i = threadIdx.x; i < C * H; i += 256) {
        const int c = i / H, k = i % H;
        lds[L::wL + k * C + c] = mlp.w[NH][i];
    }
    for (int i = threadIdx.x; i < C; i += 256) lds[L::bL + i] = mlp.b[NH][i];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hf = lane >> 5;
    float* stage = stage_all + wave * (T * 32 * C);
    int* ids = ids_all + wave * (T * 2 * 32);

    const int64_t n_macro = (E + 32 * T - 1) / (32 * T);
    for (int64_t mt = (int64_t)blockIdx.x * 4 + wave; mt < n_macro; mt += (int64_t)gridDim.x * 4) {
        const int64_t base = mt * 32 * T;
        // MLP input k = 2i+hf of this lane's edge: [y.x y.y y.z x.x x.y x.z][2i+hf], i = 0..2
        float bin[T][3];
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
        // ---- layer 0: 6 -> H ---------------------------------------------------------------------
        f32x16 h[T][KB];
#pragma unroll
        for (int ob = 0; ob < KB; ++ob) {
            f32x16 bias;
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = lds[L::b0 + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
            for (int t = 0; t < T; ++t) h[t][ob] = bias;
#pragma unroll
            for (int i = 0; i < IN0 / 2; ++i) {
                const float a = lds[L::w0 + (2 * i + hf) * H + 32 * ob + l31];
#pragma unroll
                for (int t = 0; t < T; ++t)
                    h[t][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bin[t][i], h[t][ob], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][ob][r] = gelu_fast(h[t][ob][r]);
        }
        // ---- hidden layers H -> H ----------------------------------------------------------------
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            const float* wt = lds + L::wl + (l - 1) * (H * H + H);
            f32x16 hn[T][KB];
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
                f32x16 bias;
#pragma unroll
                for (int r = 0; r < 16; ++r) bias[r] = wt[H * H + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
                for (int t = 0; t < T; ++t) hn[t][ob] = bias;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float a = wt[(32 * kb + mfma32_row(r, hf)) * H + 32 * ob + l31];
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            hn[t][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[t][kb][r], hn[t][ob], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int ob = 0; ob < KB; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r) h[t][ob][r] = gelu_fast(hn[t][ob][r]);
        }
        // ---- last layer, transposed: K'[e][c] = sum_k Hlast[k][e] * WL[c][k] + bL[c] ---------------
        wave_lds_fence();  // ids visible to the whole wave
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x16 acc;
            const float bl = lds[L::bL + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bl;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float b = lds[L::wL + (32 * kb + mfma32_row(r, hf)) * C + l31];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h[t][kb][r], b, acc, 0, 0, 0);
                }
            // multiply by the gathered feature row and stage [e][c]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int el = mfma32_row(r, hf);
                const int s = ids[(t * 2 + 0) * 32 + el];
                const float fv = f_y[(int64_t)s * C + l31];
                stage[(t * 32 + el) * C + l31] = acc[r] * fv;
            }
        }
        wave_lds_fence();
        // ---- segmented mean: half h walks tile h (T==2) / lower half walks the tile (T==1) --------
        if (hf < T) {
            segment_walk<C>(stage + hf * 32 * C, C, ids + (hf * 2 + 1) * 32, l31, base + 32 * hf, rowptr, out, part,
                            true);
        }
        wave_lds_fence();
    }
}

// =================================================================================================
// Backward
// =================================================================================================
// gs[q][:] = grad_out[q][:] / deg(q): folds the mean's 1/deg into ONE streaming pass instead of two rowptr loads
// and a divide per edge inside the backward kernel
__global__ void k_scale_by_inv_deg(const float* __restrict__ gout, const int* __restrict__ rowptr, int64_t Q,
                                   float* __restrict__ gs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index, 8 per row
    if (i >= Q * 8) return;
    const int64_t q = i >> 3;
    const int deg = rowptr[q + 1] - rowptr[q];
    const float inv = deg > 0 ? 1.0f / (float)deg : 0.f;
    float4 v = reinterpret_cast<const float4*>(gout)[i];
    reinterpret_cast<float4*>(gs)[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

// transpose the MLP weights into [in][out] (global workspace) for the recompute chain
__global__ void k_transpose_w(const float* __restrict__ w, int out_dim, int in_dim, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_dim * in_dim) return;
    const int j = i / in_dim, k = i % in_dim;
    wt[k * out_dim + j] = w[i];
}

template <int NH, int H>
struct ParamLayout {  // flat per-wave partial layout, state_dict order
    static constexpr int C = 32;
    static constexpr int w_off(int l) { return l == 0 ? 0 : (H * IN0 + H) + (l - 1) * (H * H + H); }
    static constexpr int b_off(int l) { return w_off(l) + (l == 0 ? H * IN0 : (l == NH ? C * H : H * H)); }
    static constexpr int total = (H * IN0 + H) + (NH - 1) * (H * H + H) + (C * H + C);
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// The backward kernel streams its MLP weights from global/L2 (LDS is full of activations).  The
// weight fetches are cut into groups of 16 (one per MFMA chain) in program order; group g+1 is
// issued while chain g runs (two 16-register buffers), pinned by sched_barriers so that the
// scheduler cannot hoist a whole layer of loads at once (which overflows the register file).
//   R(l,ob,kb)  recompute, hidden layer l=1..NH-1     : Wt_l[k = 32kb+row][j = 32ob+lane]
//   P(kb)       last layer, transposed product        : Wt_L[k = 32kb+row][c = lane]
//   DL(kb)      dh_NH = W_L^T dk                      : W_L[c = 2i+hf][k = 32kb+lane]
//   D(l,kb,jb)  dh_l  = W_l^T dz, l = NH-1..1         : W_l[j = 32jb+row][k = 32kb+lane]
// Weights are fetched with buffer loads: ONE per-lane voffset register per access pattern plus a
// compile-time scalar offset.  (With plain global loads every one of the ~130 loads per layer has
// its own loop-invariant 64-bit address, LICM hoists them all out of the persistent loop and the
// register file overflows by hundreds of VGPRs.)
struct WRsrc {
    __amdgpu_buffer_rsrc_t w[GAOT_MAX_MLP_LAYERS];   // W_l   [out][in]
    __amdgpu_buffer_rsrc_t wt[GAOT_MAX_MLP_LAYERS];  // W_l^T [in][out]
};
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int G, int NH, int H>
__device__ __forceinline__ void load_wgroup(float (&w)[16], const WRsrc& rs, int vo_row4 /* (4hf*H+l31)*4 */,
                                            int vo_row4c /* (4hf*32+l31)*4 */, int vo_row1 /* (hf*H+l31)*4 */) {
    constexpr int NR = 4 * (NH - 1);
    if constexpr (G < NR) {
        constexpr int l = 1 + G / 4, ob = (G % 4) / 2, kb = G % 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = bload(rs.wt[l], vo_row4, ((32 * kb + (r & 3) + 8 * (r >> 2)) * H + 32 * ob) * 4);
    } else if constexpr (G < NR + 2) {
        constexpr int kb = G - NR;
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = bload(rs.wt[NH], vo_row4c, ((32 * kb + (r & 3) + 8 * (r >> 2)) * 32) * 4);
    } else if constexpr (G < NR + 4) {
        constexpr int kb = G - NR - 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = bload(rs.w[NH], vo_row1, (2 * i * H + 32 * kb) * 4);
    } else if constexpr (G < 2 * NR + 4) {
        constexpr int i4 = G - NR - 4;
        constexpr int l = NH - 1 - i4 / 4, kb = (i4 % 4) / 2, jb = i4 % 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = bload(rs.w[l], vo_row4, ((32 * jb + (r & 3) + 8 * (r >> 2)) * H + 32 * kb) * 4);
    }
}

template <int NH, int H>
__global__ __launch_bounds__(256, 1) void k_gno_bwd(
    MlpPtrs mlp, MlpPtrs mlp_t /* w = transposed copies [in][out] */, const float* __restrict__ y_pos,
    const float* __restrict__ x_pos, const float* __restrict__ f_y, const float* __restrict__ gs /* grad_out / deg */,
    const int* __restrict__ src_s, const int* __restrict__ dst_s,
    const int* __restrict__ rowptr_src, int64_t E, float* __restrict__ grad_f, float* __restrict__ part,
    float* __restrict__ wpart /* [n_blocks][ParamLayout::total] */) {
    // One workgroup = 4 waves = 4 tiles of 32 source-sorted edges per iteration.  Each wave runs the
    // data path (recompute, dk, data gradients) of its own tile in registers; the weight-gradient
    // MFMAs are split by OUTPUT tile across the four waves (each wave sweeps all four edge tiles
    // staged in LDS), so a wave carries 16 accumulator registers per layer instead of 64.
    constexpr int C = 32;
    constexpr int KB = H / 32;
    static_assert(KB == 2, "weight-gradient wave split is written for hidden == 64");
    constexpr int LDH = H + 1;
    constexpr int NR = 4 * (NH - 1);
    using PL = ParamLayout<NH, H>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int per_wave = NH * 32 * LDH + 32 * LDH + 32 * IN0P + 64;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hf = lane >> 5;
    float* hst = lds + wave * per_wave;        // [NH][32][LDH]   h_1 .. h_NH, [edge][feature]
    float* buf = hst + NH * 32 * LDH;          // [32][LDH]       m' / dk / dz staging
    float* inst = buf + 32 * LDH;              // [32][IN0P]
    int* ids = (int*)(inst + 32 * IN0P);       // [2][32]
    float* bias_l = lds + 4 * per_wave;        // [NH][H] + [C]
    auto hst_of = [&](int w) { return lds + w * per_wave; };
    auto buf_of = [&](int w) { return lds + w * per_wave + NH * 32 * LDH; };
    auto inst_of = [&](int w) { return lds + w * per_wave + NH * 32 * LDH + 32 * LDH; };

#pragma unroll
    for (int l = 0; l < NH; ++l)
        for (int i = threadIdx.x; i < H; i += 256) bias_l[l * H + i] = mlp.b[l][i];
    for (int i = threadIdx.x; i < C; i += 256) bias_l[NH * H + i] = mlp.b[NH][i];
    __syncthreads();

    WRsrc rs;
#pragma unroll
    for (int l = 0; l <= NH; ++l) {
        const int nbytes = ((l == NH ? C : H) * (l == 0 ? IN0 : H)) * 4;
        rs.w[l] = __builtin_amdgcn_make_buffer_rsrc((void*)mlp.w[l], 0, nbytes, 0x00020000);
        rs.wt[l] = __builtin_amdgcn_make_buffer_rsrc((void*)mlp_t.w[l], 0, nbytes, 0x00020000);
    }
    const int vo_row4 = (4 * hf * H + l31) * 4, vo_row4c = (4 * hf * 32 + l31) * 4, vo_row1 = (hf * H + l31) * 4;

    // this wave's share of the weight gradients
    const int wjb = wave >> 1, wkb = wave & 1;   // hidden layers: output tile (jb,kb)
    const int pair = wave >> 1;                  // first/last layer: output tile (wave&1), edge tiles {2*pair, 2*pair+1}
    f32x16 dWL, dW0, dWh[NH > 1 ? NH - 1 : 1];
    float dbL = 0.f, db0 = 0.f, dbh[NH > 1 ? NH - 1 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dWL[r] = 0.f; dW0[r] = 0.f; }
#pragma unroll
    for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l) {
        dbh[l] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dWh[l][r] = 0.f;
    }

    float wA[16], wB[16];  // weight-group double buffer: even groups -> wA, odd -> wB

    const int64_t n_tiles = (E + 31) / 32;
    for (int64_t tb = (int64_t)blockIdx.x * 4; tb < n_tiles; tb += (int64_t)gridDim.x * 4) {
        const int64_t base = (tb + wave) * 32;
        load_wgroup<0, NH, H>(wA, rs, vo_row4, vo_row4c, vo_row1);
        // ---- gather -------------------------------------------------------------------------------
        float bin[3];  // MLP input k = 2i+hf of this lane's edge
        {
            const int64_t e = base + l31;
            const bool valid = e < E;
            const int s = valid ? src_s[e] : 0;
            const int q = valid ? dst_s[e] : 0;
            const float* ys = y_pos + (int64_t)s * 3;
            const float* xq = x_pos + (int64_t)q * 3;
            bin[0] = ys[hf];
            bin[1] = hf ? xq[0] : ys[2];
            bin[2] = xq[1 + hf];
            if (hf == 0) {
                ids[l31] = valid ? s : -1;
                ids[32 + l31] = valid ? q : -1;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) inst[l31 * IN0P + 2 * i + hf] = bin[i];
            inst[l31 * IN0P + 6 + hf] = 0.f;
        }
        // ---- recompute the MLP, keeping gelu'(z) in registers and h in LDS -------------------------
        f32x16 gp[NH][KB];
        f32x16 h[KB];
#pragma unroll
        for (int ob = 0; ob < KB; ++ob) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = bias_l[32 * ob + mfma32_row(r, hf)];
#pragma unroll
            for (int i = 0; i < IN0 / 2; ++i) {
                const float a = bload(rs.wt[0], vo_row1, (2 * i * H + 32 * ob) * 4);
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bin[i], z, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float gv, dv;
                gelu_fast_pair(z[r], gv, dv);
                h[ob][r] = gv;
                gp[0][ob][r] = dv;
                hst[(0 * 32 + l31) * LDH + 32 * ob + mfma32_row(r, hf)] = h[ob][r];
            }
        }
        static_for<1, NH>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            f32x16 z[KB];
            static_for<0, KB * KB>([&](auto gc) {
                constexpr int gi = decltype(gc)::value;
                constexpr int ob = gi / KB, kb = gi % KB;
                constexpr int G = 4 * (l - 1) + gi;
                if constexpr (kb == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[ob][r] = bias_l[l * H + 32 * ob + mfma32_row(r, hf)];
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (G % 2 == 0) load_wgroup<G + 1, NH, H>(wB, rs, vo_row4, vo_row4c, vo_row1);
                else load_wgroup<G + 1, NH, H>(wA, rs, vo_row4, vo_row4c, vo_row1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    z[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32((G % 2 == 0) ? wA[r] : wB[r], h[kb][r], z[ob], 0, 0, 0);
            });
#pragma unroll
            for (int ob = 0; ob < KB; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float gv, dv;
                    gelu_fast_pair(z[ob][r], gv, dv);
                    h[ob][r] = gv;
                    gp[l][ob][r] = dv;
                    hst[(l * 32 + l31) * LDH + 32 * ob + mfma32_row(r, hf)] = h[ob][r];
                }
        });
        // ---- gather f[src] and g[dst] rows now: their latency hides under the last layer's MFMA chain ---------
        wave_lds_fence();
        float fv[16], gv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int el = mfma32_row(r, hf);
            const int s_ = ids[el], q_ = ids[32 + el];
            const bool ok = q_ >= 0;
            gv[r] = ok ? gs[(int64_t)q_ * C + l31] : 0.f;
            fv[r] = ok ? f_y[(int64_t)s_ * C + l31] : 0.f;
        }
        // ---- last layer transposed: K'[e][c] -------------------------------------------------------
        f32x16 kp;
        {
            const float bl = bias_l[NH * H + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) kp[r] = bl;
        }
        static_for<0, KB>([&](auto kc) {
            constexpr int kb = decltype(kc)::value;
            constexpr int G = NR + kb;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G % 2 == 0) load_wgroup<G + 1, NH, H>(wB, rs, vo_row4, vo_row4c, vo_row1);
            else load_wgroup<G + 1, NH, H>(wA, rs, vo_row4, vo_row4c, vo_row1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                kp = __builtin_amdgcn_mfma_f32_32x32x2f32(h[kb][r], (G % 2 == 0) ? wA[r] : wB[r], kp, 0, 0, 0);
        });
        // ---- per edge: g = grad_out[dst]/deg (pre-scaled) ; m' = g*k' (-> grad_f) ; dk' = g*f --------------
        f32x16 dkp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int el = mfma32_row(r, hf);
            buf[el * LDH + l31] = gv[r] * kp[r];
            dkp[r] = gv[r] * fv[r];
        }
        wave_lds_fence();
        if (hf == 0) segment_walk<C>(buf, LDH, ids, l31, base, rowptr_src, grad_f, part, false);
        wave_lds_fence();
        // ---- dk' -> LDS [e][c] (own buf) ------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[mfma32_row(r, hf) * LDH + l31] = dkp[r];
        __syncthreads();
        // dW_L[c][k] += sum_e dk[c][e] * h_NH[k][e] : this wave = k-block (wave&1), edge tiles of its pair
#pragma unroll 1
        for (int tt = 0; tt < 2; ++tt) {
            const float* bt = buf_of(2 * pair + tt);
            const float* ht = hst_of(2 * pair + tt) + (NH - 1) * 32 * LDH;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const float a = bt[(2 * i + hf) * LDH + l31];
                const float b = ht[(2 * i + hf) * LDH + 32 * wkb + l31];
                if (wkb == 0) dbL += a;
                dWL = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dWL, 0, 0, 0);
            }
        }
        // own tile: dh_NH[k][e] = sum_c W_L[c][k] dk[c][e] ; dz = dh * gelu'(z_NH)
        f32x16 dz[KB];
        static_for<0, KB>([&](auto kc) {
            constexpr int kb = decltype(kc)::value;
            constexpr int G = NR + 2 + kb;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G % 2 == 0) load_wgroup<G + 1, NH, H>(wB, rs, vo_row4, vo_row4c, vo_row1);
            else load_wgroup<G + 1, NH, H>(wA, rs, vo_row4, vo_row4c, vo_row1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const float b = buf[l31 * LDH + 2 * i + hf];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32((G % 2 == 0) ? wA[i] : wB[i], b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[kb][r] = acc[r] * gp[NH - 1][kb][r];
        });
        __syncthreads();
        // ---- hidden layers, top down ----------------------------------------------------------------
        static_for<0, NH>([&](auto lrev) {
            constexpr int l = NH - 1 - decltype(lrev)::value;
            // dz (= dL/dz_{l+1}, [H][e]) -> own LDS buf as [e][feature]
#pragma unroll
            for (int jb = 0; jb < KB; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) buf[l31 * LDH + 32 * jb + mfma32_row(r, hf)] = dz[jb][r];
            __syncthreads();
            if constexpr (l > 0) {
                // dW_l[j][k] += sum_e dz[j][e] h_l[k][e] : this wave = output tile (wjb, wkb), all 4 edge tiles
#pragma unroll 1
                for (int t = 0; t < 4; ++t) {
                    const float* bt = buf_of(t);
                    const float* ht = hst_of(t) + (l - 1) * 32 * LDH;
#pragma unroll 4
                    for (int i = 0; i < 16; ++i) {
                        const float a = bt[(2 * i + hf) * LDH + 32 * wjb + l31];
                        const float b = ht[(2 * i + hf) * LDH + 32 * wkb + l31];
                        if (wkb == 0) dbh[l - 1] += a;
                        dWh[l - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dWh[l - 1], 0, 0, 0);
                    }
                }
                // own tile: dh_l[k][e] = sum_j W_l[j][k] dz[j][e] ; dz_l = dh_l * gelu'(z_l)
                f32x16 dn[KB];
                static_for<0, KB * KB>([&](auto gc) {
                    constexpr int gi = decltype(gc)::value;
                    constexpr int kb = gi / KB, jb = gi % KB;
                    constexpr int G = NR + 4 + 4 * (NH - 1 - l) + gi;
                    if constexpr (jb == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dn[kb][r] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (G % 2 == 0) load_wgroup<G + 1, NH, H>(wB, rs, vo_row4, vo_row4c, vo_row1);
                    else load_wgroup<G + 1, NH, H>(wA, rs, vo_row4, vo_row4c, vo_row1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dn[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32((G % 2 == 0) ? wA[r] : wB[r], dz[jb][r], dn[kb], 0, 0, 0);
                });
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz[kb][r] = dn[kb][r] * gp[l - 1][kb][r];
            } else {
                // dW_0[j][k<6] += sum_e dz[j][e] in[k][e] : this wave = j-block (wave&1), edge tiles of its pair
#pragma unroll 1
                for (int tt = 0; tt < 2; ++tt) {
                    const float* bt = buf_of(2 * pair + tt);
                    const float* it = inst_of(2 * pair + tt);
#pragma unroll 4
                    for (int i = 0; i < 16; ++i) {
                        const float a = bt[(2 * i + hf) * LDH + 32 * wkb + l31];
                        const float b = (l31 < IN0P) ? it[(2 * i + hf) * IN0P + l31] : 0.f;
                        db0 += a;
                        dW0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dW0, 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        });
    }

    // ---- combine the two edge-halves of dW_L / dW_0 (waves 2,3 -> waves 0,1), then write the block partial
    __syncthreads();
    float* xch = lds;  // [2 waves][2 tiles][16][64] + bias scalars
    if (wave >= 2) {
        float* x = xch + (wave - 2) * (2 * 16 * 64 + 2 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r * 64 + lane] = dWL[r];
            x[(16 + r) * 64 + lane] = dW0[r];
        }
        x[32 * 64 + lane] = dbL;
        x[33 * 64 + lane] = db0;
    }
    __syncthreads();
    float* wp = wpart + (int64_t)blockIdx.x * PL::total;
    if (wave < 2) {
        const float* x = xch + wave * (2 * 16 * 64 + 2 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dWL[r] += x[r * 64 + lane];
            dW0[r] += x[(16 + r) * 64 + lane];
        }
        dbL += x[32 * 64 + lane];
        db0 += x[33 * 64 + lane];
        // dW_L tile: rows c = row(r,hf), cols k = 32*wave + l31
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[PL::w_off(NH) + mfma32_row(r, hf) * H + 32 * wave + l31] = dWL[r];
        // dW_0 tile: rows j = 32*wave + row(r,hf), cols k = l31 (<6)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (l31 < IN0) wp[PL::w_off(0) + (32 * wave + mfma32_row(r, hf)) * IN0 + l31] = dW0[r];
        const float t0 = db0 + __shfl_xor(db0, 32, 64);
        if (hf == 0) wp[PL::b_off(0) + 32 * wave + l31] = t0;
        if (wave == 0) {
            const float tl = dbL + __shfl_xor(dbL, 32, 64);
            if (hf == 0) wp[PL::b_off(NH) + l31] = tl;
        }
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            wp[PL::w_off(l) + (32 * wjb + mfma32_row(r, hf)) * H + 32 * wkb + l31] = dWh[l - 1][r];
        if (wkb == 0) {
            const float t = dbh[l - 1] + __shfl_xor(dbh[l - 1], 32, 64);
            if (hf == 0) wp[PL::b_off(l) + 32 * wjb + l31] = t;
        }
    }
}

// grad[p] = sum over waves (fixed order) of the per-wave partials; scattered to the per-tensor outputs
// (a workgroup owns 64 parameters; its 4 waves each sum every 4th partial, then the 4 sums are added in wave order)
__global__ __launch_bounds__(256) void k_reduce_params(const float* __restrict__ wpart, int n_waves, int total,
                                                       float* __restrict__ flat) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + o;
    float s = 0.f;
    if (p < total)
        for (int w = sl; w < n_waves; w += 4) s += wpart[(int64_t)w * total + p];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && p < total) flat[p] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

struct ScatterDesc {
    float* dst[2 * GAOT_MAX_MLP_LAYERS];
    int off[2 * GAOT_MAX_MLP_LAYERS + 1];
    int n;
};
__global__ void k_scatter_params(const float* __restrict__ flat, ScatterDesc d) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.off[d.n]) return;
    int t = 0;
    while (p >= d.off[t + 1]) ++t;
    d.dst[t][p - d.off[t]] = flat[p];
}

int param_total(int nh, int h) { return (h * IN0 + h) + (nh - 1) * (h * h + h) + (32 * h + 32); }

size_t fwd_lds_bytes(int nh, int h, int t) {
    const int weights = IN0P * h + h + (nh - 1) * (h * h + h) + h * 32 + 32;
    return sizeof(float) * (size_t)(weights + 4 * t * 32 * 32) + sizeof(int) * (size_t)(4 * t * 2 * 32);
}
size_t bwd_lds_bytes(int nh, int h) {
    const int per_wave = nh * 32 * (h + 1) + 32 * (h + 1) + 32 * IN0P + 64;
    return sizeof(float) * (size_t)(4 * per_wave + nh * h + 32);
}

template <int NH, int H>
int launch_fwd(const MlpPtrs& p, const float* y_pos, const float* x_pos, const float* f_y, const int* src_s,
               const int* dst_s, const int* rowptr, int64_t E, float* out, float* part, hipStream_t st) {
    constexpr int T = 2;
    const size_t lds = fwd_lds_bytes(NH, H, T);
    auto kern = k_gno_fwd<NH, H, T>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        gaot_set_error("gno_fwd: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
        return GAOT_ERR_LAUNCH;
    }
    const int64_t n_macro = ceil_div(E, 32 * T);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_macro, 4), 256 * 2));
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part);
    return GAOT_OK;
}

template <int NH, int H>
int launch_bwd(const MlpPtrs& p, const MlpPtrs& pt, const float* y_pos, const float* x_pos, const float* f_y,
               const float* gs, const int* src_s, const int* dst_s, const int* rowptr_src,
               int64_t E, float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    const size_t lds = bwd_lds_bytes(NH, H);
    auto kern = k_gno_bwd<NH, H>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        gaot_set_error("gno_bwd: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
        return GAOT_ERR_LAUNCH;
    }
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, p, pt, y_pos, x_pos, f_y, gs, src_s, dst_s,
                       rowptr_src, E, grad_f, part, wpart);
    return GAOT_OK;
}

int bwd_grid(int64_t E) {
    const int64_t n_tiles = ceil_div(E, 32);
    return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_tiles, 4), 256));
}

bool mlp_supported(const gaot_mlp_t* m, bool backward, int precision = 0) {
    // the exact-fp32 backward keeps h_1..h_NH of four 32-edge tiles in LDS: NH = 4 would need 172 KB (> 160 KB/CU); the bf16
    // backward reads its operand fragments from L2 at four hidden layers (gno_bwd3_bf16.hip) and takes them
    return m && m->channels == 32 && (m->hidden == 64) && m->n_hidden >= 1 &&
           m->n_hidden <= (backward ? (precision == 1 ? 4 : 3) : 4);
}

}  // namespace

int gaot_gno_fwd_bf16_dispatch(int n_hidden, const float* const* w, const float* const* b, const float* y_pos,
                               const float* x_pos, const float* f_y, const int32_t* src_sorted, const int32_t* dst_sorted,
                               const int32_t* rowptr_dst, int64_t num_edges, float* out, float* part, hipStream_t st);

size_t gaot_gno_bwd_bf16_image_bytes(int n_hidden);
int gaot_gno_bwd_bf16_dispatch(int n_hidden, const float* const* w, const float* const* b, const float* w0t,
                               void* images, const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                               const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                               int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st);

extern "C" size_t gaot_gno_fwd_workspace_bytes(int64_t num_edges, int channels) {
    return sizeof(float) * (size_t)(ceil_div(num_edges, 32) * 2 * channels) + 64;
}

extern "C" int gaot_gno_fwd(const gaot_mlp_t* mlp, const float* y_pos, const float* x_pos, const float* f_y,
                            const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_dst,
                            int64_t num_edges, int64_t num_queries, float* out, int precision, void* workspace,
                            size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(mlp, "null mlp");
    if (!mlp_supported(mlp, false)) {
        gaot_set_error("gaot_gno_fwd: unsupported MLP shape (n_hidden=%d hidden=%d channels=%d)", mlp->n_hidden,
                       mlp->hidden, mlp->channels);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(num_edges >= 0 && num_queries >= 0, "negative size");
    GAOT_CHECK_ARG(rowptr_dst && (num_queries == 0 || out), "null pointer");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_gno_fwd_workspace_bytes(num_edges, mlp->channels), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (num_queries == 0) return GAOT_OK;
    MlpPtrs p;
    for (int l = 0; l <= mlp->n_hidden; ++l) {
        p.w[l] = mlp->weight[l];
        p.b[l] = mlp->bias[l];
        GAOT_CHECK_ARG(p.w[l] && p.b[l], "null MLP parameter");
    }
    float* part = (float*)workspace;
    int rc = GAOT_OK;
    if (num_edges > 0) {
        GAOT_CHECK_ARG(y_pos && x_pos && f_y && src_sorted && dst_sorted, "null pointer");
        GAOT_CHECK_ARG(precision == 0 || precision == 1, "precision must be 0 (fp32) or 1 (bf16 matrix cores)");
        if (precision == 1) {
            rc = gaot_gno_fwd_bf16_dispatch(mlp->n_hidden, p.w, p.b, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst,
                                            num_edges, out, part, st);
        } else
        switch (mlp->n_hidden) {
            case 1: rc = launch_fwd<1, 64>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st); break;
            case 2: rc = launch_fwd<2, 64>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st); break;
            case 3: rc = launch_fwd<3, 64>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st); break;
            case 4: rc = launch_fwd<4, 64>(p, y_pos, x_pos, f_y, src_sorted, dst_sorted, rowptr_dst, num_edges, out, part, st); break;
        }
        if (rc != GAOT_OK) return rc;
    }
    const int64_t n = num_queries * 8;   // four channels per thread
    GAOT_KLAUNCH((k_segment_fixup<32>), dim3(segment_fixup_grid(n)), dim3(256), 0, st, rowptr_dst,
                       num_queries, part, out, 1);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" size_t gaot_gno_bwd_workspace_bytes(const gaot_mlp_t* mlp, int64_t num_edges, int64_t num_queries) {
    if (!mlp) return 0;
    const int total = param_total(mlp->n_hidden, mlp->hidden);
    const int grid = bwd_grid(num_edges);
    size_t fl = (size_t)(ceil_div(num_edges, 16) * 2 * 32)  // segment partials (16-edge tiles in bf16 mode, 32 in fp32)
                + (size_t)total                             // transposed weights (<= total)
                + (size_t)total                             // reduced flat gradient
                + (size_t)grid * total                      // per-block partials
                + (size_t)num_queries * 32;                 // grad_out / deg
    return sizeof(float) * fl + gaot_gno_bwd_bf16_image_bytes(mlp->n_hidden) + 512;
}

extern "C" int gaot_gno_bwd(const gaot_mlp_t* mlp, const float* y_pos, const float* x_pos, const float* f_y,
                            const float* grad_out, const int32_t* rowptr_dst, const int32_t* src_sorted,
                            const int32_t* dst_sorted, const int32_t* rowptr_src, int64_t num_edges,
                            int64_t num_sources, int64_t num_queries, float* grad_f_y, const gaot_mlp_grad_t* grads,
                            int precision, void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(mlp && grads, "null mlp");
    if (!mlp_supported(mlp, true, precision)) {
        gaot_set_error("gaot_gno_bwd: unsupported MLP shape (n_hidden=%d hidden=%d channels=%d; four hidden layers in bf16 mode only)", mlp->n_hidden,
                       mlp->hidden, mlp->channels);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(num_edges >= 0 && num_sources >= 0 && num_queries >= 0, "negative size");
    // the bf16 kernel gathers its 128-byte rows through 2 GB buffer resources: 2^24 rows per table
    GAOT_CHECK_ARG(precision != 1 || (num_sources <= (1 << 24) && num_queries <= (1 << 24)),
                   "bf16 GNO backward: more than 2^24 source or query rows (run the mesh point-sharded or in fp32 mode)");
    // ... and addresses the two 128-byte partial slots of a 16-edge tile with a 32-bit byte offset
    GAOT_CHECK_ARG(precision != 1 || num_edges < ((int64_t)1 << 27),
                   "bf16 GNO backward: 2^27 or more edges in one launch (run the mesh point-sharded or in fp32 mode)");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_gno_bwd_workspace_bytes(mlp, num_edges, num_queries), "workspace too small");
    GAOT_CHECK_ARG(rowptr_src && rowptr_dst, "null rowptr");
    hipStream_t st = (hipStream_t)stream;
    const int nh = mlp->n_hidden, h = mlp->hidden;
    const int total = param_total(nh, h);
    const int grid = bwd_grid(num_edges);
    const int n_waves = grid;  // one partial per workgroup
    float* part = (float*)workspace;
    float* wt = part + ceil_div(num_edges, 16) * 2 * 32;
    float* flat = wt + total;
    float* wpart = flat + total;
    float* gs = wpart + (size_t)grid * total;
    void* images = (void*)(((uintptr_t)(gs + (size_t)num_queries * 32) + 255) & ~(uintptr_t)255);
    GAOT_CHECK_ARG(precision == 0 || precision == 1, "precision must be 0 (fp32) or 1 (bf16 matrix cores)");

    MlpPtrs p, pt;
    int off = 0;
    for (int l = 0; l <= nh; ++l) {
        p.w[l] = mlp->weight[l];
        p.b[l] = mlp->bias[l];
        GAOT_CHECK_ARG(p.w[l] && p.b[l] && grads->weight[l] && grads->bias[l], "null MLP parameter / gradient");
        const int out_dim = (l == nh) ? 32 : h;
        const int in_dim = (l == 0) ? IN0 : h;
        pt.w[l] = wt + off;
        pt.b[l] = p.b[l];
        if (num_edges > 0)
            GAOT_KLAUNCH(k_transpose_w, dim3((unsigned)ceil_div(out_dim * in_dim, 256)), dim3(256), 0, st, p.w[l],
                               out_dim, in_dim, wt + off);
        off += out_dim * in_dim;
    }
    ScatterDesc sd;
    sd.n = 2 * (nh + 1);
    {
        int o = 0;
        for (int l = 0; l <= nh; ++l) {
            const int out_dim = (l == nh) ? 32 : h;
            const int in_dim = (l == 0) ? IN0 : h;
            sd.dst[2 * l] = grads->weight[l];
            sd.off[2 * l] = o;
            o += out_dim * in_dim;
            sd.dst[2 * l + 1] = grads->bias[l];
            sd.off[2 * l + 1] = o;
            o += out_dim;
        }
        sd.off[sd.n] = o;
    }
    if (num_edges == 0) {
        hipMemsetAsync(flat, 0, sizeof(float) * total, st);
    } else {
        GAOT_CHECK_ARG(y_pos && x_pos && f_y && grad_out && src_sorted && dst_sorted && grad_f_y, "null pointer");
        GAOT_KLAUNCH(k_scale_by_inv_deg, dim3((unsigned)ceil_div(num_queries * 8, 256)), dim3(256), 0, st, grad_out,
                           rowptr_dst, num_queries, gs);
        int rc = GAOT_OK;
        if (precision == 1) {
            rc = gaot_gno_bwd_bf16_dispatch(nh, p.w, p.b, pt.w[0], images, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted,
                                            rowptr_src, num_edges, grad_f_y, part, wpart, grid, st);
        } else
        switch (nh) {
            case 1: rc = launch_bwd<1, 64>(p, pt, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f_y, part, wpart, grid, st); break;
            case 2: rc = launch_bwd<2, 64>(p, pt, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f_y, part, wpart, grid, st); break;
            case 3: rc = launch_bwd<3, 64>(p, pt, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f_y, part, wpart, grid, st); break;
        }
        if (rc != GAOT_OK) return rc;
        GAOT_KLAUNCH(k_reduce_params, dim3((unsigned)ceil_div(total, 64)), dim3(256), 0, st, wpart, n_waves,
                           total, flat);
    }
    GAOT_KLAUNCH(k_scatter_params, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, flat, sd);
    if (num_sources > 0) {
        const int64_t n = num_sources * 8;   // four channels per thread
        if (precision == 1)
            GAOT_KLAUNCH((k_segment_fixup<32, 4>), dim3(segment_fixup_grid(n)), dim3(256), 0, st, rowptr_src,
                               num_sources, part, grad_f_y, 0);
        else
            GAOT_KLAUNCH((k_segment_fixup<32, 5>), dim3(segment_fixup_grid(n)), dim3(256), 0, st, rowptr_src,
                               num_sources, part, grad_f_y, 0);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
