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

template <int NH, int T>
__global__ __launch_bounds__(256, 2) void k_gno_fwd_bf16(MlpPtrs mlp, const float* __restrict__ y_pos,
                                                         const float* __restrict__ x_pos, const float* __restrict__ f_y,
                                                         const int* __restrict__ src_s, const int* __restrict__ dst_s,
                                                         const int* __restrict__ rowptr, int64_t E, float* __restrict__ out,
                                                         float* __restrict__ part) {
    constexpr int C = 32, H = 64, KB = 2;
    using L = FwdLdsB<NH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage_all = lds + L::weights_end;             // [4 waves][T][32][C]
    int* ids_all = (int*)(stage_all + 4 * T * 32 * C);   // [4 waves][T][2][32]

    // ---- layer 0 weights (fp32, transposed) and all biases ------------------------------------------
    for (int i = threadIdx.x; i < IN0P * H; i += 256) {
        const int k = i / H, j = i % H;
        lds[L::w0 + i] = (k < IN0) ? mlp.w[0][j * IN0 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < H; i += 256) lds[L::b0 + i] = mlp.b[0][i];
#pragma unroll
    for (int l = 1; l < NH; ++l)
        for (int i = threadIdx.x; i < H; i += 256) lds[L::bl + (l - 1) * H + i] = mlp.b[l][i];
    for (int i = threadIdx.x; i < C; i += 256) lds[L::bL + i] = mlp.b[NH][i];
    // ---- bf16 fragment images: hidden layer l, fragment (ob,kb,s), lane, element j ---------------------
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        bf16_t* im = reinterpret_cast<bf16_t*>(lds + L::img + (l - 1) * L::img_hidden_floats);
        for (int i = threadIdx.x; i < KB * KB * 2 * 64 * 8; i += 256) {
            const int j = i & 7, ln = (i >> 3) & 63, fr = i >> 9;   // fr = (ob*KB + kb)*2 + s
            const int s = fr & 1, kb = (fr >> 1) % KB, ob = (fr >> 1) / KB;
            const int row = 32 * ob + (ln & 31), col = 32 * kb + kmap(s, j, ln >> 5);
            im[i] = (bf16_t)f2bf(mlp.w[l][row * H + col]);
        }
    }
    {
        bf16_t* im = reinterpret_cast<bf16_t*>(lds + L::imgL);
        for (int i = threadIdx.x; i < KB * 2 * 64 * 8; i += 256) {
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
    for (int64_t mt = (int64_t)blockIdx.x * 4 + wave; mt < n_macro; mt += (int64_t)gridDim.x * 4) {
        const int64_t base = mt * 32 * T;
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
                    const f32v2 y = gelu_fast2(f32v2{z[t][r], z[t][r + 1]});
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
                        const f32v2 y = gelu_fast2(f32v2{z[t][ob][r], z[t][ob][r + 1]});
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
                const int s = ids[(t * 2 + 0) * 32 + el];
                const float fv = f_y[(int64_t)s * C + l31];
                stage[(t * 32 + el) * C + l31] = acc[r] * fv;
            }
        }
        wave_lds_fence();
        if (hf < T) {
            segment_walk<C>(stage + hf * 32 * C, C, ids + (hf * 2 + 1) * 32, l31, base + 32 * hf, rowptr, out, part, true);
        }
        wave_lds_fence();
    }
}

size_t fwd_lds_bytes_b(int nh, int t) {
    const int weights = IN0P * 64 + 64 + (nh - 1) * 64 + 32 + (nh - 1) * (2 * 2 * 2 * 64 * 8 / 2) + (2 * 2 * 64 * 8 / 2);
    return sizeof(float) * (size_t)(weights + 4 * t * 32 * 32) + sizeof(int) * (size_t)(4 * t * 2 * 32);
}

template <int NH>
int launch_fwd_b(const MlpPtrs& p, const float* y_pos, const float* x_pos, const float* f_y, const int* src_s,
                 const int* dst_s, const int* rowptr, int64_t E, float* out, float* part, hipStream_t st) {
    constexpr int T = 2;
    const size_t lds = fwd_lds_bytes_b(NH, T);
    auto kern = k_gno_fwd_bf16<NH, T>;
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
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_macro, 4), 256 * 2));
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, p, y_pos, x_pos, f_y, src_s, dst_s, rowptr, E, out, part);
    return GAOT_OK;
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
// Backward (bf16 matrix cores)
// =================================================================================================
// Same decomposition as k_gno_bwd (gno.hip): one workgroup = 4 waves = 4 tiles of 32 source-sorted edges per
// iteration; every wave runs the data path of its own tile in registers, the weight-gradient products are split by
// OUTPUT tile across the waves and sweep all four edge tiles staged in LDS.  Differences:
//   * activations h_l, dz_l and dk are staged as bf16 [feature][edge] tiles (64-B rows, 16-B chunks XOR-swizzled by
//     (row>>2)&3): a weight-gradient operand is one ds_read_b128 along the edge axis;
//   * bias gradients ride on the matrix cores too: dz (or dk) times a one-hot "selector" fragment drops the edge-sum of
//     layer l into column l of ONE shared accumulator tile; db_0 comes out of column 6 of the dW_0 tile (a row of ones
//     in the staged input tile);
//   * the MLP weights are pre-arranged (k_prep_bwd_images) as per-lane bf16 fragments in global memory and fetched with
//     one 16-byte buffer load per MFMA.
namespace {

constexpr int TILE_B = 64 * 64;   // bytes of a [64 feat][32 e] bf16 tile
__device__ __forceinline__ int t_addr(int row, int e) { return row * 64 + ((((e >> 3) ^ ((row >> 2) & 3))) << 4) + ((e & 7) << 1); }
// operand contracting over the edge axis (contiguous): lane (row, hf), k-step s -> edges 16s + 8hf + j
__device__ __forceinline__ bf16x8 t_frag(const char* tile, int row, int s, int hf) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((((2 * s + hf) ^ ((row >> 2) & 3))) << 4));
}
// operand contracting over tile ROWS (transposed read): lane = column e, element j <-> row kmap(s, j, hf)
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 t_frag_tr(const char* tile, int lane, int s) {
    const int i = lane & 15, grp = (lane >> 4) & 1, hf = lane >> 5;
    const int col = 16 * grp + 4 * (i & 3);
    const int r0 = 16 * s + 4 * hf + (i >> 2), r1 = r0 + 8;
    const char* p0 = tile + r0 * 64 + ((((col >> 3) ^ ((r0 >> 2) & 3))) << 4) + ((col & 7) << 1);
    const char* p1 = tile + r1 * 64 + ((((col >> 3) ^ ((r1 >> 2) & 3))) << 4) + ((col & 7) << 1);
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}
// store a D-layout fp32 tile block (lane = edge l31, reg r <-> feature fbase + row(r,hf)) as bf16 [feature][edge]
__device__ __forceinline__ void t_store_feat(char* tile, int fbase, const f32x16& v, int l31, int hf) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<short*>(tile + t_addr(fbase + mfma32_row(r, hf), l31)) = f2bf(v[r]);
}

struct BwdImgs {
    const bf16_t* fw[GAOT_MAX_MLP_LAYERS];   // recompute: layer l fragments (ob,kb,s) ; fw[NH] = last layer (kb,s)
    const bf16_t* bw[GAOT_MAX_MLP_LAYERS];   // data-grad: layer l fragments (kb,jb,s) of W_l^T ; bw[NH] = (kb,s) of W_L^T
    const float* w0t;                        // fp32 [6][64]
};

// fragment images: see the operand maps in the kernel
__global__ void k_prep_bwd_images(MlpPtrs mlp, int nh, bf16_t* base) {
    constexpr int H = 64, KB = 2;
    const int per_hidden = KB * KB * 2 * 64 * 8, per_last = KB * 2 * 64 * 8;
    const int total = (nh - 1) * 2 * per_hidden + 2 * per_last;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int o = i;
        float v;
        if (o < (nh - 1) * per_hidden) {                       // fw[l], l = 1..nh-1
            const int l = 1 + o / per_hidden; o %= per_hidden;
            const int j = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, kb = (fr >> 1) % KB, ob = (fr >> 1) / KB;
            v = mlp.w[l][(32 * ob + (ln & 31)) * H + 32 * kb + kmap(s, j, ln >> 5)];
        } else if ((o -= (nh - 1) * per_hidden) < per_last) {  // fw[nh]: B operand of the transposed last layer
            const int j = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, kb = fr >> 1;
            v = mlp.w[nh][(ln & 31) * H + 32 * kb + kmap(s, j, ln >> 5)];
        } else if ((o -= per_last) < (nh - 1) * per_hidden) {  // bw[l]: A = W_l^T, rows k, elements j
            const int l = 1 + o / per_hidden; o %= per_hidden;
            const int j = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, jb = (fr >> 1) % KB, kb = (fr >> 1) / KB;
            v = mlp.w[l][(32 * jb + kmap(s, j, ln >> 5)) * H + 32 * kb + (ln & 31)];
        } else {                                               // bw[nh]: A = W_L^T, rows k, elements c
            o -= (nh - 1) * per_hidden;
            const int j = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, kb = fr >> 1;
            v = mlp.w[nh][kmap(s, j, ln >> 5) * H + 32 * kb + (ln & 31)];
        }
        base[i] = (bf16_t)f2bf(v);
    }
}

template <int NH>
struct ParamLayoutB {  // flat per-workgroup partial layout, state_dict order (same as gno.hip)
    static constexpr int H = 64, C = 32;
    static constexpr int w_off(int l) { return l == 0 ? 0 : (H * IN0 + H) + (l - 1) * (H * H + H); }
    static constexpr int b_off(int l) { return w_off(l) + (l == 0 ? H * IN0 : (l == NH ? C * H : H * H)); }
    static constexpr int total = (H * IN0 + H) + (NH - 1) * (H * H + H) + (C * H + C);
};

__device__ __forceinline__ bf16x8 bload8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int NH>
__global__ __launch_bounds__(256, 1) void k_gno_bwd_bf16(
    BwdImgs im, MlpPtrs mlp, const float* __restrict__ y_pos, const float* __restrict__ x_pos,
    const float* __restrict__ f_y, const float* __restrict__ gs, const int* __restrict__ src_s,
    const int* __restrict__ dst_s, const int* __restrict__ rowptr_src, int64_t E, float* __restrict__ grad_f,
    float* __restrict__ part, float* __restrict__ wpart) {
    constexpr int C = 32, H = 64, KB = 2;
    using PL = ParamLayoutB<NH>;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    constexpr int per_wave = NH * TILE_B + TILE_B + 2048 + 512 + 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    auto hT_of = [&](int w, int l) { return ldsb + w * per_wave + l * TILE_B; };
    auto dzT_of = [&](int w) { return ldsb + w * per_wave + NH * TILE_B; };
    auto dkT_of = [&](int w) { return ldsb + w * per_wave + NH * TILE_B + TILE_B; };
    auto inT_of = [&](int w) { return ldsb + w * per_wave + NH * TILE_B + TILE_B + 2048; };
    char* dzT = dzT_of(wave);
    char* dkT = dkT_of(wave);
    char* inT = inT_of(wave);
    float* mbuf = reinterpret_cast<float*>(dzT);               // m' tile [32 e][32 c] fp32 aliases the dz tile
    int* ids = reinterpret_cast<int*>(inT + 512);
    float* bias_l = reinterpret_cast<float*>(ldsb + 4 * per_wave);   // [NH][H] + [C]

#pragma unroll
    for (int l = 0; l < NH; ++l)
        for (int i = threadIdx.x; i < H; i += 256) bias_l[l * H + i] = mlp.b[l][i];
    for (int i = threadIdx.x; i < C; i += 256) bias_l[NH * H + i] = mlp.b[NH][i];
    __syncthreads();

    __amdgpu_buffer_rsrc_t rfw[NH + 1], rbw[NH + 1];
#pragma unroll
    for (int l = 1; l <= NH; ++l) {
        const int nbytes = (l == NH ? KB * 2 : KB * KB * 2) * 64 * 16;
        rfw[l] = __builtin_amdgcn_make_buffer_rsrc((void*)im.fw[l], 0, nbytes, 0x00020000);
        rbw[l] = __builtin_amdgcn_make_buffer_rsrc((void*)im.bw[l], 0, nbytes, 0x00020000);
    }
    __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)im.w0t, 0, IN0 * H * 4, 0x00020000);
    const int vo16 = lane * 16;
    const int vo_w0 = (hf * H + l31) * 4;

    const int wjb = wave >> 1, wkb = wave & 1, pair = wave >> 1;
    f32x16 dWL, dW0, dWh[NH > 1 ? NH - 1 : 1], bacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dWL[r] = 0.f; dW0[r] = 0.f; bacc[r] = 0.f; }
#pragma unroll
    for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWh[l][r] = 0.f;
    auto sel = [&](int col) {   // one-hot selector fragment: B[e][n] = (n == col)
        bf16x8 f;
        const short one = (l31 == col) ? (short)0x3F80 : (short)0;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = one;
        return f;
    };

    const int64_t n_tiles = (E + 31) / 32;
    // The tile's edge ids and endpoint coordinates are two DEPENDENT global round trips (ids, then y[src] / x[dst]);
    // with one wave per SIMD nothing else hides them, so they are fetched one iteration ahead.
    int s_nx = 0, q_nx = 0;
    bool v_nx = false;
    float bin_nx[3] = {0.f, 0.f, 0.f};
    auto fetch_ids = [&](int64_t tb_) {
        const int64_t e = (tb_ + wave) * 32 + l31;
        v_nx = tb_ < n_tiles && e < E;
        s_nx = v_nx ? src_s[e] : 0;
        q_nx = v_nx ? dst_s[e] : 0;
    };
    auto fetch_pos = [&]() {
        const float* ys = y_pos + (int64_t)s_nx * 3;
        const float* xq = x_pos + (int64_t)q_nx * 3;
        bin_nx[0] = ys[hf];
        bin_nx[1] = hf ? xq[0] : ys[2];
        bin_nx[2] = xq[1 + hf];
    };
    fetch_ids((int64_t)blockIdx.x * 4);
    fetch_pos();
    for (int64_t tb = (int64_t)blockIdx.x * 4; tb < n_tiles; tb += (int64_t)gridDim.x * 4) {
        const int64_t base = (tb + wave) * 32;
        // ---- indices and coordinates of this tile (prefetched); start the next tile's id fetch ------------------
        float bin[3];
        {
            const bool valid = v_nx;
            const int s = s_nx, q = q_nx;
            bin[0] = bin_nx[0]; bin[1] = bin_nx[1]; bin[2] = bin_nx[2];
            fetch_ids(tb + (int64_t)gridDim.x * 4);
            if (hf == 0) {
                ids[l31] = valid ? s : -1;
                ids[32 + l31] = valid ? q : -1;
            }
            // input tile [k][e] bf16 (rows 0..5 = coordinates, row 6 = ones -> db_0, row 7 = 0)
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<short*>(inT + t_addr(2 * i + hf, l31)) = f2bf(bin[i]);
            *reinterpret_cast<short*>(inT + t_addr(6 + hf, l31)) = hf ? (short)0 : (short)0x3F80;
        }
        // ---- recompute the MLP: gelu'(z) stays in registers (fp32), h_l goes to LDS as bf16 ---------------------
        f32x16 gp[NH][KB];
        bf16x8 hb[KB][2];
#pragma unroll
        for (int ob = 0; ob < KB; ++ob) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = bias_l[32 * ob + mfma32_row(r, hf)];
#pragma unroll
            for (int i = 0; i < IN0 / 2; ++i) {
                const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw0, vo_w0, (2 * i * H + 32 * ob) * 4, 0));
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bin[i], z, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float gv_, dv_;
                gelu_fast_pair(z[r], gv_, dv_);
                z[r] = gv_;
                gp[0][ob][r] = dv_;
            }
            t_store_feat(hT_of(wave, 0), 32 * ob, z, l31, hf);
            to_frags(z, hb[ob][0], hb[ob][1]);
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            f32x16 z[KB];
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[ob][r] = bias_l[l * H + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        z[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bload8(rfw[l], vo16, (((ob * KB + kb) * 2 + s) * 64) * 16),
                                                                        hb[kb][s], z[ob], 0, 0, 0);
            }
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float gv_, dv_;
                    gelu_fast_pair(z[ob][r], gv_, dv_);
                    z[ob][r] = gv_;
                    gp[l][ob][r] = dv_;
                }
                t_store_feat(hT_of(wave, l), 32 * ob, z[ob], l31, hf);
                to_frags(z[ob], hb[ob][0], hb[ob][1]);
            }
        }
        fetch_pos();   // next tile's coordinates: its ids were requested at the top of this tile
        // ---- gather f[src] / g[dst] rows (latency hides under the last layer) ----------------------------------
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
        // ---- last layer transposed: K'[e][c] ---------------------------------------------------------------------
        f32x16 kp;
        {
            const float blv = bias_l[NH * H + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) kp[r] = blv;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    kp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb[kb][s], bload8(rfw[NH], vo16, ((kb * 2 + s) * 64) * 16), kp, 0, 0, 0);
        }
        // ---- m' = g*k' -> grad_f (segmented sum over the source-sorted tile); dk' = g*f --------------------------
        f32x16 dkp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mbuf[mfma32_row(r, hf) * C + l31] = gv[r] * kp[r];
            dkp[r] = gv[r] * fv[r];
        }
        wave_lds_fence();
        if (hf == 0) segment_walk<C>(mbuf, C, ids, l31, base, rowptr_src, grad_f, part, false);
        wave_lds_fence();
        // ---- dk' -> LDS tile [c][e] (lane = channel row, 4 runs of 4 consecutive edges) ---------------------------
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint2 pk = make_uint2((unsigned)(unsigned short)f2bf(dkp[4 * g4]) | ((unsigned)(unsigned short)f2bf(dkp[4 * g4 + 1]) << 16),
                                        (unsigned)(unsigned short)f2bf(dkp[4 * g4 + 2]) | ((unsigned)(unsigned short)f2bf(dkp[4 * g4 + 3]) << 16));
            *reinterpret_cast<uint2*>(dkT + t_addr(l31, 8 * g4 + 4 * hf)) = pk;
        }
        __syncthreads();
        // dW_L[c][k] += sum_e dk[c][e] h_NH[k][e]: this wave = k-block wkb, edge tiles of its pair
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const char* dt = dkT_of(2 * pair + tt);
            const char* ht = hT_of(2 * pair + tt, NH - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                dWL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t_frag(dt, l31, s, hf), t_frag(ht, 32 * wkb + l31, s, hf), dWL, 0, 0, 0);
        }
        if (wave == 0) {   // db_L[c] = sum_e dk[c][e] -> column NH of the bias tile
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t_frag(dkT_of(t), l31, s, hf), sel(NH), bacc, 0, 0, 0);
        }
        // own tile: dh_NH[k][e] = sum_c W_L[c][k] dk[c][e] ; dz = dh * gelu'(z_NH)
        f32x16 dz[KB];
        {
            const bf16x8 dk0 = t_frag_tr(dkT, lane, 0), dk1 = t_frag_tr(dkT, lane, 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bload8(rbw[NH], vo16, ((kb * 2 + 0) * 64) * 16), dk0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bload8(rbw[NH], vo16, ((kb * 2 + 1) * 64) * 16), dk1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[kb][r] = acc[r] * gp[NH - 1][kb][r];
            }
        }
        __syncthreads();
        // ---- hidden layers, top down ---------------------------------------------------------------------------------
#pragma unroll
        for (int l = NH - 1; l >= 0; --l) {
            bf16x8 dzb[KB][2];
#pragma unroll
            for (int jb = 0; jb < KB; ++jb) {
                t_store_feat(dzT, 32 * jb, dz[jb], l31, hf);
                to_frags(dz[jb], dzb[jb][0], dzb[jb][1]);
            }
            __syncthreads();
            if (l > 0) {
                // dW_l[j][k] += sum_e dz[j][e] h_l[k][e]: this wave = output tile (wjb, wkb), all 4 edge tiles
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const char* zt = dzT_of(t);
                    const char* ht = hT_of(t, l - 1);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 a = t_frag(zt, 32 * wjb + l31, s, hf);
                        dWh[l - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, t_frag(ht, 32 * wkb + l31, s, hf), dWh[l - 1], 0, 0, 0);
                        if (wkb == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(l), bacc, 0, 0, 0);
                    }
                }
                // own tile: dh_l[k][e] = sum_j W_l[j][k] dz[j][e] ; dz_l = dh_l * gelu'(z_l)
                f32x16 dn[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dn[kb][r] = 0.f;
#pragma unroll
                    for (int jb = 0; jb < KB; ++jb)
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            dn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bload8(rbw[l], vo16, (((kb * KB + jb) * 2 + s) * 64) * 16),
                                                                             dzb[jb][s], dn[kb], 0, 0, 0);
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz[kb][r] = dn[kb][r] * gp[l - 1][kb][r];
            } else {
                // dW_0[j][k] += sum_e dz[j][e] in[k][e] (k = 6 is the ones row -> db_0): j-block wkb, edge tiles of its pair
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const char* zt = dzT_of(2 * pair + tt);
                    const char* it = inT_of(2 * pair + tt);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 b = t_frag(it, l31 & 7, s, hf);
                        if (l31 >= 8) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) b[j] = 0;
                        }
                        dW0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t_frag(zt, 32 * wkb + l31, s, hf), b, dW0, 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- combine the two edge-halves of dW_L / dW_0 (waves 2,3 -> waves 0,1), then write the workgroup partial ----------
    __syncthreads();
    float* xch = reinterpret_cast<float*>(ldsb);
    if (wave >= 2) {
        float* x = xch + (wave - 2) * (2 * 16 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r * 64 + lane] = dWL[r];
            x[(16 + r) * 64 + lane] = dW0[r];
        }
    }
    __syncthreads();
    float* wp = wpart + (int64_t)blockIdx.x * PL::total;
    if (wave < 2) {
        const float* x = xch + wave * (2 * 16 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dWL[r] += x[r * 64 + lane];
            dW0[r] += x[(16 + r) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[PL::w_off(NH) + mfma32_row(r, hf) * H + 32 * wave + l31] = dWL[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * wave + mfma32_row(r, hf);
            if (l31 < IN0) wp[PL::w_off(0) + j * IN0 + l31] = dW0[r];
            if (l31 == 6) wp[PL::b_off(0) + j] = dW0[r];
        }
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            wp[PL::w_off(l) + (32 * wjb + mfma32_row(r, hf)) * H + 32 * wkb + l31] = dWh[l - 1][r];
        if (wkb == 0 && l31 == l) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[PL::b_off(l) + 32 * wjb + mfma32_row(r, hf)] = bacc[r];
        }
    }
    if (wave == 0 && l31 == NH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[PL::b_off(NH) + mfma32_row(r, hf)] = bacc[r];
    }
}

size_t bwd_lds_bytes_b(int nh) { return (size_t)4 * (nh * TILE_B + TILE_B + 2048 + 512 + 256) + sizeof(float) * (nh * 64 + 32); }
int bwd_img_elems(int nh) { return (nh - 1) * 2 * (2 * 2 * 2 * 64 * 8) + 2 * (2 * 2 * 64 * 8); }

template <int NH>
int launch_bwd_b(const BwdImgs& im, const MlpPtrs& p, const float* y_pos, const float* x_pos, const float* f_y,
                 const float* gs, const int* src_s, const int* dst_s, const int* rowptr_src, int64_t E, float* grad_f,
                 float* part, float* wpart, int grid, hipStream_t st) {
    const size_t lds = bwd_lds_bytes_b(NH);
    auto kern = k_gno_bwd_bf16<NH>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            gaot_set_error("gno_bwd_bf16: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, im, p, y_pos, x_pos, f_y, gs, src_s, dst_s, rowptr_src, E,
                       grad_f, part, wpart);
    return GAOT_OK;
}

}  // namespace

int gaot_gno_bwd2_bf16_launch(int n_hidden, const void* images, const float* w0t, const float* const* w, const float* const* b,
                              const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                              const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                              int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st);

size_t gaot_gno_bwd_bf16_image_bytes(int n_hidden) { return sizeof(bf16_t) * (size_t)bwd_img_elems(n_hidden) + 256; }

// called from gaot_gno_bwd (gno.hip) for precision == 1.  `images` = scratch of gaot_gno_bwd_bf16_image_bytes();
// w0t = fp32 [6][64] transposed first-layer weight (prepared by the caller, shared with the fp32 path).
int gaot_gno_bwd_bf16_dispatch(int n_hidden, const float* const* w, const float* const* b, const float* w0t,
                               void* images, const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                               const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                               int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    MlpPtrs p;
    for (int l = 0; l <= n_hidden; ++l) { p.w[l] = w[l]; p.b[l] = b[l]; }
    bf16_t* base = (bf16_t*)images;
    GAOT_KLAUNCH(k_prep_bwd_images, dim3(32), dim3(256), 0, st, p, n_hidden, base);
    const int per_hidden = 2 * 2 * 2 * 64 * 8, per_last = 2 * 2 * 64 * 8;
    BwdImgs im{};
    bf16_t* q = base;
    for (int l = 1; l < n_hidden; ++l) { im.fw[l] = q; q += per_hidden; }
    im.fw[n_hidden] = q; q += per_last;
    for (int l = 1; l < n_hidden; ++l) { im.bw[l] = q; q += per_hidden; }
    im.bw[n_hidden] = q;
    im.w0t = w0t;
    // second design (gno_bwd2_bf16.hip: LDS-resident operand images, one weight-gradient phase per 128 edges);
    // GAOT_GNO_BWD_V1=1 keeps the first design for A/B measurements
    static const bool v1 = getenv("GAOT_GNO_BWD_V1") != nullptr;
    if (!v1)
        return gaot_gno_bwd2_bf16_launch(n_hidden, images, w0t, w, b, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src,
                                         num_edges, grad_f, part, wpart, grid, st);
    switch (n_hidden) {
        case 1: return launch_bwd_b<1>(im, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 2: return launch_bwd_b<2>(im, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 3: return launch_bwd_b<3>(im, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
    }
    gaot_set_error("gaot_gno_bwd (bf16): unsupported n_hidden %d", n_hidden);
    return GAOT_ERR_UNSUPPORTED;
}
