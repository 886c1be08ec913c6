// Fused per-node two-layer MLP  out = W2 gelu(W1 x + b1) + b2  (reference MAGNODecoder.projection: C -> 256 -> out
// with GELU, src/model/layers/magno.py:793-797 -> LinearChannelMLP/ChannelMLP, mlp.py:227-335) and its autograd,
// bf16 matrix-core products with fp32 accumulation (precision 1).  The unfused chain writes and re-reads a
// [N, 256] fp32 hidden tensor 8 times per step (N = 500 000: 512 MB each); here the hidden layer never leaves
// registers: forward reads x (128 B/row) and writes out; backward recomputes the hidden layer, reads x and d_out
// and writes d_x plus per-workgroup weight-gradient partials (fixed-order reduction, bit-reproducible).
//
// One workgroup = 128 rows, hidden units split over its 4 waves.  x is staged once per tile as 32 x 32 bf16 LDS
// tiles (tile32.h).  Two accumulator orientations are used, each the B operand of what follows it:
//   P: z[j][n] (hidden on registers, row on lanes)  -> out[c][n] and d_x^T[k][n]  (contract over hidden)
//   Q: z^T[n][j] (row on registers, hidden on lanes) -> d_W1^T[k][j], d_W2, d_b1    (contract over rows)
// The backward computes dz ONCE, in the Q orientation (GELU / GELU' share one exp: gelu_e2_pair, common.h), and hands its bf16
// rounding to the d_x product through a wave-private [hidden][row] LDS tile read back transposed (ds_read_b64_tr_b16) -- the
// round-3 kernel recomputed z and GELU' in the P orientation (2 MFMAs, 8 more exponentials and 16 OC fused multiply-adds per
// 32 x 32 block); the K = 32 product was cheap, the second activation pass was not.
#include "common.h"
#include "tile32.h"

namespace {

constexpr int MLP_IN = 32;
constexpr int MLP_ROWS = 128;   // rows per workgroup tile

struct Mlp2Args {
    const float* x; int64_t N;
    const float* w1; const float* b1;   // [H, 32], [H]
    const float* w2; const float* b2;   // [OC, H], [OC]
};

// stage 128 rows of x (fp32 [N,32]) as four 32x32 bf16 tiles; rows >= N are zero
__device__ __forceinline__ void stage_x(const float* __restrict__ x, int64_t row0, int64_t N, char* xt) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int chunk = threadIdx.x + it * 256;    // 512 chunks of 8 elements
        const int row = chunk >> 2, c = chunk & 3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (row0 + row < N) {
            const float* p = x + (row0 + row) * MLP_IN + 8 * c;
            a = *reinterpret_cast<const float4*>(p);
            b = *reinterpret_cast<const float4*>(p + 4);
        }
        uint4 pk;
        pk.x = (unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
        pk.y = (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
        pk.z = (unsigned)f2bf(b.x) | ((unsigned)f2bf(b.y) << 16);
        pk.w = (unsigned)f2bf(b.z) | ((unsigned)f2bf(b.w) << 16);
        *reinterpret_cast<uint4*>(xt + (row >> 5) * TILE_BYTES + tile_off(row & 31, c)) = pk;
    }
}

// W1 fragment with the hidden unit on the lane: rows 32*ob + l31, k = 16s + 8hf + j
__device__ __forceinline__ bf16x8 w1_frag(const float* __restrict__ w1, int ob, int s, int l31, int hf) {
    const float* p = w1 + (int64_t)(32 * ob + l31) * MLP_IN + 16 * s + 8 * hf;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 f;
    f[0] = (short)f2bf(a.x); f[1] = (short)f2bf(a.y); f[2] = (short)f2bf(a.z); f[3] = (short)f2bf(a.w);
    f[4] = (short)f2bf(b.x); f[5] = (short)f2bf(b.y); f[6] = (short)f2bf(b.z); f[7] = (short)f2bf(b.w);
    return f;
}

template <int NOB, int OC>
__global__ __launch_bounds__(256, 2) void k_mlp2_fwd(Mlp2Args a, float* __restrict__ out) {
    constexpr int H = 32 * NOB, OBW = NOB >= 4 ? NOB / 4 : 1;
    __shared__ __attribute__((aligned(16))) char xt[4 * TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float b1s[H];
    __shared__ __attribute__((aligned(16))) float w2s[OC * H];
    __shared__ float part[4][MLP_ROWS][OC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    for (int i = threadIdx.x; i < H; i += 256) b1s[i] = a.b1[i];
    for (int i = threadIdx.x; i < OC * H; i += 256) w2s[i] = a.w2[i];
    bf16x8 w1f[OBW][2];
#pragma unroll
    for (int o = 0; o < OBW; ++o) {
        const int ob = wave * OBW + o;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (ob < NOB) w1f[o][s] = w1_frag(a.w1, ob, s, l31, hf);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) w1f[o][s][j] = 0;
            }
        }
    }
    for (int64_t row0 = (int64_t)blockIdx.x * MLP_ROWS; row0 < a.N; row0 += (int64_t)gridDim.x * MLP_ROWS) {
        __syncthreads();
        stage_x(a.x, row0, a.N, xt);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float po[OC];
#pragma unroll
            for (int c = 0; c < OC; ++c) po[c] = 0.f;
#pragma unroll
            for (int o = 0; o < OBW; ++o) {
                const int ob = wave * OBW + o;
                if (ob >= NOB) continue;
                f32x16 z;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * ob + 8 * g4 + 4 * hf]);
                    z[4 * g4] = bv.x; z[4 * g4 + 1] = bv.y; z[4 * g4 + 2] = bv.z; z[4 * g4 + 3] = bv.w;
                }
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[o][0], frag_rows(xt + t * TILE_BYTES, l31, hf, 0), z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[o][1], frag_rows(xt + t * TILE_BYTES, l31, hf, 1), z, 0, 0, 0);
#pragma unroll
                for (int r2 = 0; r2 < 16; r2 += 2) {   // two activations per packed-fp32 instruction
                    const f32v2 y2 = gelu_e2_2(f32v2{z[r2], z[r2 + 1]});
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int c = 0; c < OC; ++c) po[c] = fmaf(y2[u], w2s[c * H + 32 * ob + mfma32_row(r2 + u, hf)], po[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < OC; ++c) {
                const float v = po[c] + __shfl_xor(po[c], 32, 64);
                if (hf == 0) part[wave][32 * t + l31][c] = v;
            }
        }
        __syncthreads();
        if (threadIdx.x < MLP_ROWS && row0 + threadIdx.x < a.N) {
#pragma unroll
            for (int c = 0; c < OC; ++c) {
                float v = a.b2 ? a.b2[c] : 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (NOB >= 4 || w < NOB) v += part[w][threadIdx.x][c];
                out[(row0 + threadIdx.x) * OC + c] = v;
            }
        }
    }
}

// per-workgroup partial layout: dW1^T? no -- dW1 [H][32], then db1 [H], then dW2 [OC][H]
template <int NOB, int OC>
__global__ __launch_bounds__(256, 1) void k_mlp2_bwd(Mlp2Args a, const float* __restrict__ dout, float* __restrict__ dx,
                                                     float* __restrict__ wpart) {
    constexpr int H = 32 * NOB, OBW = NOB >= 4 ? NOB / 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // > 64 KB: dynamic (bwd_lds_bytes)
    char* xt = smem;                                                        // 4 x-tiles
    char* w1t = xt + 4 * TILE_BYTES;                                        // W1 as bf16 tiles [ob][32 hidden][32 feature]
    float* b1s = reinterpret_cast<float*>(w1t + NOB * TILE_BYTES);          // [H]
    float* dos = b1s + H;                                                   // d_out tile [row][c]
    float (*dxp)[32][MLP_ROWS + 1] = reinterpret_cast<float (*)[32][MLP_ROWS + 1]>(dos + MLP_ROWS * OC);  // per-wave d_x^T partials
    char* dzt_all = reinterpret_cast<char*>(dxp + 4);                       // per wave: dz tile [32 hidden][32 rows] bf16
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    for (int i = threadIdx.x; i < H; i += 256) b1s[i] = a.b1[i];
    char* dzt = dzt_all + wave * TILE_BYTES;
    for (int ch = threadIdx.x; ch < H * 4; ch += 256) {   // W1 -> LDS tiles, 8 elements per chunk
        const int j = ch >> 2, c = ch & 3;
        const float* p = a.w1 + (int64_t)j * MLP_IN + 8 * c;
        const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + 4);
        uint4 pk;
        pk.x = (unsigned)f2bf(u.x) | ((unsigned)f2bf(u.y) << 16);
        pk.y = (unsigned)f2bf(u.z) | ((unsigned)f2bf(u.w) << 16);
        pk.z = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
        pk.w = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
        *reinterpret_cast<uint4*>(w1t + (j >> 5) * TILE_BYTES + tile_off(j & 31, c)) = pk;
    }
    bf16x8 w1f[OBW][2];
    float b1l[OBW], w2l[OBW][OC];        // per-lane (hidden = 32*ob + l31) scalars of the Q orientation
    f32x16 dw1t[OBW];                     // d_W1^T tiles [feature k][hidden j]
    float db1[OBW], dw2[OBW][OC];
#pragma unroll
    for (int o = 0; o < OBW; ++o) {
        const int ob = wave * OBW + o;
        const bool on = ob < NOB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (on) w1f[o][s] = w1_frag(a.w1, ob, s, l31, hf);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) w1f[o][s][j] = 0;
            }
        }
        b1l[o] = on ? a.b1[32 * ob + l31] : 0.f;
        db1[o] = 0.f;
#pragma unroll
        for (int c = 0; c < OC; ++c) { w2l[o][c] = on ? a.w2[c * H + 32 * ob + l31] : 0.f; dw2[o][c] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) dw1t[o][r] = 0.f;
    }
    for (int64_t row0 = (int64_t)blockIdx.x * MLP_ROWS; row0 < a.N; row0 += (int64_t)gridDim.x * MLP_ROWS) {
        __syncthreads();
        stage_x(a.x, row0, a.N, xt);
        for (int i = threadIdx.x; i < MLP_ROWS * OC; i += 256) {
            const int64_t row = row0 + i / OC;
            dos[i] = row < a.N ? dout[row * OC + i % OC] : 0.f;   // rows >= N contribute nothing
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char* xtile = xt + t * TILE_BYTES;
            f32x16 dxt;   // d_x^T [feature][row] of this 32-row sub-tile, over this wave's hidden units
#pragma unroll
            for (int r = 0; r < 16; ++r) dxt[r] = 0.f;
            // d_out of the 16 rows of this lane's accumulator registers (rows 32 t + mfma32_row(r, hf)): read once per sub-tile,
            // shared by the wave's hidden blocks (the round-3 kernel re-read them from LDS inside every block)
            float dor[16][OC];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < OC; ++c) dor[r][c] = dos[(32 * t + mfma32_row(r, hf)) * OC + c];
#pragma unroll
            for (int o = 0; o < OBW; ++o) {
                const int ob = wave * OBW + o;
                if (ob >= NOB) continue;
                const bf16x8 xr0 = frag_rows(xtile, l31, hf, 0), xr1 = frag_rows(xtile, l31, hf, 1);
                // ---- z^T[n][j] (row on registers, hidden on lanes) -> dz -> d_W1^T, d_W2, d_b1 -------------------------
                f32x16 zq;
#pragma unroll
                for (int r = 0; r < 16; ++r) zq[r] = b1l[o];
                zq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr0, w1f[o][0], zq, 0, 0, 0);
                zq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr1, w1f[o][1], zq, 0, 0, 0);
                float s1 = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < 16; r2 += 2) {   // two activations per packed-fp32 instruction
                    f32v2 y2, gd2;
                    gelu_e2_pair2(f32v2{zq[r2], zq[r2 + 1]}, y2, gd2);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int r = r2 + u;
                        float dy = 0.f;
#pragma unroll
                        for (int c = 0; c < OC; ++c) {
                            dy = fmaf(dor[r][c], w2l[o][c], dy);
                            dw2[o][c] = fmaf(dor[r][c], y2[u], dw2[o][c]);
                        }
                        const float dz = dy * gd2[u];
                        s1 += dz;
                        zq[r] = dz;
                    }
                }
                db1[o] += s1;
                bf16x8 f0, f1;
                acc_to_frags(zq, f0, f1);
                dw1t[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(xtile, lane, 0), f0, dw1t[o], 0, 0, 0);
                dw1t[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(xtile, lane, 1), f1, dw1t[o], 0, 0, 0);
                // ---- d_x^T[k][n] += sum_j W1[j][k] dz[n][j]: the same bf16 dz, as the tile [hidden = l31][row] (the lane's four runs
                //      of four consecutive rows = 8-byte stores), read back transposed: lane = row, elements = hidden in the
                //      accumulator-as-operand order.  LDS operations of one wave complete in order.
                {
                    const uint4 lo = __builtin_bit_cast(uint4, f0), hi = __builtin_bit_cast(uint4, f1);
                    *reinterpret_cast<uint2*>(dzt + tile_off(l31, 0) + 8 * hf) = make_uint2(lo.x, lo.y);
                    *reinterpret_cast<uint2*>(dzt + tile_off(l31, 1) + 8 * hf) = make_uint2(lo.z, lo.w);
                    *reinterpret_cast<uint2*>(dzt + tile_off(l31, 2) + 8 * hf) = make_uint2(hi.x, hi.y);
                    *reinterpret_cast<uint2*>(dzt + tile_off(l31, 3) + 8 * hf) = make_uint2(hi.z, hi.w);
                    dxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(w1t + ob * TILE_BYTES, lane, 0), frag_cols(dzt, lane, 0), dxt, 0, 0, 0);
                    dxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(w1t + ob * TILE_BYTES, lane, 1), frag_cols(dzt, lane, 1), dxt, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dxp[wave][mfma32_row(r, hf)][32 * t + l31] = dxt[r];
        }
        __syncthreads();
        // d_x[row][k] = sum over waves of the partials, in wave order; thread = (row, 16 features)
        {
            const int row = threadIdx.x >> 1, k0 = (threadIdx.x & 1) * 16;
            if (row0 + row < a.N) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float s = dxp[0][k0 + k][row];
#pragma unroll
                    for (int w = 1; w < 4; ++w)
                        if (NOB >= 4 || w < NOB) s += dxp[w][k0 + k][row];
                    v[k] = s;
                }
                float* dp = dx + (row0 + row) * MLP_IN + k0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dp + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
        }
    }
    // workgroup partial: dW1 [H][32] | db1 [H] | dW2 [OC][H]
    float* wp = wpart + (int64_t)blockIdx.x * (H * MLP_IN + H + OC * H);
#pragma unroll
    for (int o = 0; o < OBW; ++o) {
        const int ob = wave * OBW + o;
        if (ob >= NOB) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[(int64_t)(32 * ob + l31) * MLP_IN + mfma32_row(r, hf)] = dw1t[o][r];   // [j][k]
        const float d1 = db1[o] + __shfl_xor(db1[o], 32, 64);
        if (hf == 0) wp[H * MLP_IN + 32 * ob + l31] = d1;
#pragma unroll
        for (int c = 0; c < OC; ++c) {
            const float d2 = dw2[o][c] + __shfl_xor(dw2[o][c], 32, 64);
            if (hf == 0) wp[H * MLP_IN + H + c * H + 32 * ob + l31] = d2;
        }
    }
}

// out[i] = sum over workgroups of part[g * stride + i], i < count: a workgroup owns 64 outputs, its 4 waves each sum
// every 4th partial, the 4 sums are added in wave order (fixed order -> bit-reproducible)
__global__ __launch_bounds__(256) void k_mlp2_reduce(const float* __restrict__ part, int groups, int64_t stride, int64_t count,
                                                     float* __restrict__ out) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < count)
        for (int g = sl; g < groups; g += 4) s += part[(int64_t)g * stride + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < count) out[i] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

constexpr size_t bwd_lds_bytes(int nob, int oc) {
    return (size_t)4 * TILE_BYTES + (size_t)nob * TILE_BYTES + sizeof(float) * (32 * nob + MLP_ROWS * oc) +
           sizeof(float) * 4 * 32 * (MLP_ROWS + 1) + (size_t)4 * TILE_BYTES;
}

constexpr int MLP_BWD_GRID = 256;

template <int NOB>
int fwd_oc(const Mlp2Args& a, int oc, float* out, hipStream_t st) {
    const int grid = (int)std::min<int64_t>(ceil_div(a.N, MLP_ROWS), 2048);
    switch (oc) {
        case 1: GAOT_KLAUNCH((k_mlp2_fwd<NOB, 1>), dim3(grid), dim3(256), 0, st, a, out); return GAOT_OK;
        case 2: GAOT_KLAUNCH((k_mlp2_fwd<NOB, 2>), dim3(grid), dim3(256), 0, st, a, out); return GAOT_OK;
        case 3: GAOT_KLAUNCH((k_mlp2_fwd<NOB, 3>), dim3(grid), dim3(256), 0, st, a, out); return GAOT_OK;
        case 4: GAOT_KLAUNCH((k_mlp2_fwd<NOB, 4>), dim3(grid), dim3(256), 0, st, a, out); return GAOT_OK;
    }
    return GAOT_ERR_UNSUPPORTED;
}
template <int NOB, int OC>
int bwd_launch(const Mlp2Args& a, const float* dout, float* dx, float* wpart, int grid, hipStream_t st) {
    constexpr size_t lds = bwd_lds_bytes(NOB, OC);
    auto kern = k_mlp2_bwd<NOB, OC>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            gaot_set_error("mlp2_bwd: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, a, dout, dx, wpart);
    return GAOT_OK;
}
template <int NOB>
int bwd_oc(const Mlp2Args& a, int oc, const float* dout, float* dx, float* wpart, int grid, hipStream_t st) {
    switch (oc) {
        case 1: return bwd_launch<NOB, 1>(a, dout, dx, wpart, grid, st);
        case 2: return bwd_launch<NOB, 2>(a, dout, dx, wpart, grid, st);
        case 3: return bwd_launch<NOB, 3>(a, dout, dx, wpart, grid, st);
        case 4: return bwd_launch<NOB, 4>(a, dout, dx, wpart, grid, st);
    }
    return GAOT_ERR_UNSUPPORTED;
}

int check_shape(int in_dim, int hidden, int out_dim) {
    if (in_dim != MLP_IN || (hidden != 64 && hidden != 128 && hidden != 256) || out_dim < 1 || out_dim > 4) {
        gaot_set_error("gaot_mlp2: supported shapes are in = 32, hidden in {64, 128, 256}, out in 1..4 (got %d -> %d -> %d)", in_dim,
                       hidden, out_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    return GAOT_OK;
}

}  // namespace

extern "C" int gaot_mlp2_fwd(const float* x, int64_t num_rows, int in_dim, int hidden, int out_dim, const float* w1,
                             const float* b1, const float* w2, const float* b2, float* out, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_rows >= 0, "negative size");
    if (int rc = check_shape(in_dim, hidden, out_dim)) return rc;
    if (num_rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(x && w1 && b1 && w2 && out, "null pointer");
    GAOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1) & 15) == 0, "x and w1 must be 16-byte aligned");
    Mlp2Args a{x, num_rows, w1, b1, w2, b2};
    hipStream_t st = (hipStream_t)stream;
    int rc = hidden == 64 ? fwd_oc<2>(a, out_dim, out, st) : (hidden == 128 ? fwd_oc<4>(a, out_dim, out, st) : fwd_oc<8>(a, out_dim, out, st));
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" size_t gaot_mlp2_bwd_workspace_bytes(int hidden, int out_dim) {
    return sizeof(float) * (size_t)MLP_BWD_GRID * (hidden * MLP_IN + hidden + out_dim * hidden) + 64;
}

extern "C" int gaot_mlp2_bwd(const float* x, int64_t num_rows, int in_dim, int hidden, int out_dim, const float* w1,
                             const float* b1, const float* w2, const float* d_out, float* d_x, float* d_w1, float* d_b1,
                             float* d_w2, void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_rows >= 0, "negative size");
    if (int rc = check_shape(in_dim, hidden, out_dim)) return rc;
    GAOT_CHECK_ARG(w1 && b1 && w2 && d_w1 && d_b1 && d_w2, "null pointer");
    GAOT_CHECK_ARG(workspace && workspace_bytes >= gaot_mlp2_bwd_workspace_bytes(hidden, out_dim), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t np = (int64_t)hidden * MLP_IN + hidden + (int64_t)out_dim * hidden;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(num_rows, MLP_ROWS), MLP_BWD_GRID));
    float* wpart = (float*)workspace;
    if (num_rows > 0) {
        GAOT_CHECK_ARG(x && d_out && d_x, "null pointer");
        GAOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)d_x) & 15) == 0, "x, w1 and d_x must be 16-byte aligned");
    }
    Mlp2Args a{x, num_rows, w1, b1, w2, nullptr};
    int rc = hidden == 64 ? bwd_oc<2>(a, out_dim, d_out, d_x, wpart, grid, st)
                          : (hidden == 128 ? bwd_oc<4>(a, out_dim, d_out, d_x, wpart, grid, st) : bwd_oc<8>(a, out_dim, d_out, d_x, wpart, grid, st));
    if (rc != GAOT_OK) return rc;
    // partial layout [dW1 | db1 | dW2] -> the three outputs (contiguous pieces of one reduction)
    const int64_t n1 = (int64_t)hidden * MLP_IN, n2 = hidden, n3 = (int64_t)out_dim * hidden;
    GAOT_KLAUNCH(k_mlp2_reduce, dim3((unsigned)ceil_div(n1, 64)), dim3(256), 0, st, wpart, grid, np, n1, d_w1);
    GAOT_KLAUNCH(k_mlp2_reduce, dim3((unsigned)ceil_div(n2, 64)), dim3(256), 0, st, wpart + n1, grid, np, n2, d_b1);
    GAOT_KLAUNCH(k_mlp2_reduce, dim3((unsigned)ceil_div(n3, 64)), dim3(256), 0, st, wpart + n1 + n2, grid, np, n3, d_w2);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
