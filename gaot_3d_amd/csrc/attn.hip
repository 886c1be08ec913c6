// Multi-head attention for the latent Transformer (reference GroupQueryFlashAttention.forward,
// src/model/layers/attn.py:110-127: head split, GQA repeat, F.scaled_dot_product_attention with no
// mask, scale 1/sqrt(head_dim), dropout 0) and its autograd -- flash-style, never materialising
// the S x S score matrix.  head_dim is fixed to 32 (hidden 256 / 8 heads in every shipped config).
//
// fp32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 products).  One wave owns 32 query rows (forward,
// dQ) or 32 key rows (dK/dV); the other side streams through LDS in 32-row tiles.  All products
// are arranged so that the accumulator tile of one MFMA chain is directly the B operand of the
// next ("reduction index on the register axis"), so P / dS never go through LDS:
//   forward : S^T[key][q] = K Q^T  ->  P^T (softmax over registers + one cross-half shuffle)
//             O^T[d][q]  += V^T[d][key] P^T[key][q]
//   dK/dV   : S[q][key], dP[q][key] with the wave's keys on lanes;  dV^T += dO^T P, dK^T += Q^T dS
//   dQ      : S^T, dP^T with the wave's queries on lanes;            dQ^T += K^T dS^T
// dQ is produced by its own pass (no float atomics: results are bit-reproducible).
#include "common.h"

namespace {

constexpr int D = 32;       // head dim
constexpr int LDP = 33;     // padded LDS row (k-contiguous reads by 32 lanes on 32 rows)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
    const float* q; const float* k; const float* v;
    float* o; float* lse;                       // lse: [B][H][S] natural log
    int64_t ldq, ldk, ldv, ldo;
    int B, S, H, HKV;
    float scale;
};

struct AttnBwdArgs {
    const float* q; const float* k; const float* v; const float* o; const float* d_o; const float* lse;
    float* delta;                                // [B][H][S]
    float* dq; float* dk; float* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, S, H, HKV;
    float scale;
};

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// cooperative load of a [32][D] fp32 tile (rows row0.., zero beyond nrows) into LDS with row pitch ld
__device__ __forceinline__ float4 tile_ld(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows) {
    const int r = threadIdx.x >> 3, c4 = threadIdx.x & 7;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows) t = *reinterpret_cast<const float4*>(base + (row0 + r) * ld + 4 * c4);
    return t;
}
__device__ __forceinline__ void tile_st(float* lds, int pitch, float4 t) {
    const int r = threadIdx.x >> 3, c4 = threadIdx.x & 7;
    float* p = lds + r * pitch + 4 * c4;
    p[0] = t.x; p[1] = t.y; p[2] = t.z; p[3] = t.w;
}

// ------------------------------------------------------------------------------------------------
// forward: block = 4 waves x 32 queries; grid (ceil(S/128), H, B)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_attn_fwd_f32(AttnArgs a) {
    __shared__ float Ks[32 * LDP];
    __shared__ float Vs[32 * LDP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const float* qp = a.q + rowbase * a.ldq + head * D;
    const float* kp = a.k + rowbase * a.ldk + hkv * D;
    const float* vp = a.v + rowbase * a.ldv + hkv * D;
    const float sc = a.scale * LOG2E;

    float qreg[16];
    {
        const int64_t qi = q0 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) qreg[i] = (qi < a.S) ? qp[qi * a.ldq + 2 * i + hf] * sc : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float m = -INFINITY, l = 0.f;

    float4 kt = tile_ld(kp, a.ldk, 0, a.S), vt = tile_ld(vp, a.ldv, 0, a.S);
    for (int64_t k0 = 0; k0 < a.S; k0 += 32) {
        __syncthreads();
        tile_st(Ks, LDP, kt);
        tile_st(Vs, LDP, vt);
        __syncthreads();
        if (k0 + 32 < a.S) {
            kt = tile_ld(kp, a.ldk, k0 + 32, a.S);
            vt = tile_ld(vp, a.ldv, k0 + 32, a.S);
        }
        // S^T[key][q]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * LDP + 2 * i + hf], qreg[i], s, 0, 0, 0);
        if (k0 + 32 > a.S) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + mfma32_row(r, hf) >= a.S) s[r] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = exp2f(s[r] - mn);
            ps += s[r];
        }
        ps += xhalf(ps);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= alpha;
        // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[mfma32_row(i, hf) * LDP + l31], s[i], acc, 0, 0, 0);
    }
    const int64_t qi = q0 + l31;
    if (qi < a.S) {
        const float inv = 1.f / l;
        float* op = a.o + (rowbase + qi) * a.ldo + head * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hf) = t;
        }
        if (hf == 0) a.lse[((int64_t)b * a.H + head) * a.S + qi] = m * LN2 + logf(l);
    }
}

// ------------------------------------------------------------------------------------------------
// delta[b][h][s] = sum_d dO * O
// ------------------------------------------------------------------------------------------------
__global__ void k_attn_delta(AttnBwdArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)a.B * a.S * a.H;
    if (i >= n) return;
    const int head = (int)(i % a.H);
    const int64_t row = i / a.H;  // b*S + s
    const float* op = a.o + row * a.ldo + head * D;
    const float* dp = a.d_o + row * a.lddo + head * D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const float4 x = *reinterpret_cast<const float4*>(op + 4 * c);
        const float4 y = *reinterpret_cast<const float4*>(dp + 4 * c);
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    const int64_t bb = row / a.S, ss = row % a.S;
    a.delta[(bb * a.H + head) * a.S + ss] = s;
}

// ------------------------------------------------------------------------------------------------
// dK / dV: block = 4 waves x 32 keys; grid (ceil(S/128), HKV, B); loops over the group's q heads
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_f32(AttnBwdArgs a) {
    __shared__ float Qs[32 * LDP];
    __shared__ float dOs[32 * LDP];
    __shared__ float lse_s[32];
    __shared__ float del_s[32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.y, b = blockIdx.z;
    const int rep = a.H / a.HKV;
    const int64_t key0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const float sc = a.scale * LOG2E;
    const int64_t ki = key0 + l31;
    float kreg[16], vreg[16];
    {
        const float* kp = a.k + (rowbase + ki) * a.ldk + hkv * D;
        const float* vp = a.v + (rowbase + ki) * a.ldv + hkv * D;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            kreg[i] = (ki < a.S) ? kp[2 * i + hf] * sc : 0.f;
            vreg[i] = (ki < a.S) ? vp[2 * i + hf] : 0.f;
        }
    }
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }

    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const float* qp = a.q + rowbase * a.ldq + head * D;
        const float* dop = a.d_o + rowbase * a.lddo + head * D;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        float4 qt = tile_ld(qp, a.ldq, 0, a.S), dt = tile_ld(dop, a.lddo, 0, a.S);
        float lt = 0.f, et = 0.f;
        if (threadIdx.x < 32) {
            lt = (threadIdx.x < a.S) ? lsep[threadIdx.x] * LOG2E : INFINITY;
            et = (threadIdx.x < a.S) ? delp[threadIdx.x] : 0.f;
        }
        for (int64_t q0 = 0; q0 < a.S; q0 += 32) {
            __syncthreads();
            tile_st(Qs, LDP, qt);
            tile_st(dOs, LDP, dt);
            if (threadIdx.x < 32) { lse_s[threadIdx.x] = lt; del_s[threadIdx.x] = et; }
            __syncthreads();
            if (q0 + 32 < a.S) {
                qt = tile_ld(qp, a.ldq, q0 + 32, a.S);
                dt = tile_ld(dop, a.lddo, q0 + 32, a.S);
                if (threadIdx.x < 32) {
                    const int64_t qq = q0 + 32 + threadIdx.x;
                    lt = (qq < a.S) ? lsep[qq] * LOG2E : INFINITY;
                    et = (qq < a.S) ? delp[qq] : 0.f;
                }
            }
            // S[q][key] ; dP[q][key]
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[l31 * LDP + 2 * i + hf], kreg[i], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[l31 * LDP + 2 * i + hf], vreg[i], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = mfma32_row(r, hf);
                const float p = exp2f(s[r] - lse_s[qr]);
                s[r] = p;                          // P
                dp[r] = p * (dp[r] - del_s[qr]);   // dS (without the 1/sqrt(d) factor)
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qr = mfma32_row(i, hf);
                dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[qr * LDP + l31], s[i], dvt, 0, 0, 0);
                dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qr * LDP + l31], dp[i], dkt, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (ki < a.S) {
        float* dkp = a.dk + (rowbase + ki) * a.lddk + hkv * D;
        float* dvp = a.dv + (rowbase + ki) * a.lddv + hkv * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(dkt[4 * g] * a.scale, dkt[4 * g + 1] * a.scale, dkt[4 * g + 2] * a.scale,
                                   dkt[4 * g + 3] * a.scale);
            *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = t;
            float4 u = make_float4(dvt[4 * g], dvt[4 * g + 1], dvt[4 * g + 2], dvt[4 * g + 3]);
            *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: block = 4 waves x 32 queries; grid (ceil(S/128), H, B)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq_f32(AttnBwdArgs a) {
    __shared__ float Ks[32 * LDP];
    __shared__ float Vs[32 * LDP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const float* kp = a.k + rowbase * a.ldk + hkv * D;
    const float* vp = a.v + rowbase * a.ldv + hkv * D;
    const float sc = a.scale * LOG2E;
    const int64_t qi = q0 + l31;
    float qreg[16], doreg[16];
    {
        const float* qp = a.q + (rowbase + qi) * a.ldq + head * D;
        const float* dop = a.d_o + (rowbase + qi) * a.lddo + head * D;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            qreg[i] = (qi < a.S) ? qp[2 * i + hf] * sc : 0.f;
            doreg[i] = (qi < a.S) ? dop[2 * i + hf] : 0.f;
        }
    }
    const float lse2 = (qi < a.S) ? a.lse[((int64_t)b * a.H + head) * a.S + qi] * LOG2E : INFINITY;
    const float del = (qi < a.S) ? a.delta[((int64_t)b * a.H + head) * a.S + qi] : 0.f;
    f32x16 dqt;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[r] = 0.f;

    float4 kt = tile_ld(kp, a.ldk, 0, a.S), vt = tile_ld(vp, a.ldv, 0, a.S);
    for (int64_t k0 = 0; k0 < a.S; k0 += 32) {
        __syncthreads();
        tile_st(Ks, LDP, kt);
        tile_st(Vs, LDP, vt);
        __syncthreads();
        if (k0 + 32 < a.S) {
            kt = tile_ld(kp, a.ldk, k0 + 32, a.S);
            vt = tile_ld(vp, a.ldv, k0 + 32, a.S);
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * LDP + 2 * i + hf], qreg[i], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[l31 * LDP + 2 * i + hf], doreg[i], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = exp2f(s[r] - lse2);
            if (k0 + mfma32_row(r, hf) >= a.S) p = 0.f;
            dp[r] = p * (dp[r] - del);
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int i = 0; i < 16; ++i)
            dqt = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[mfma32_row(i, hf) * LDP + l31], dp[i], dqt, 0, 0, 0);
    }
    if (qi < a.S) {
        float* dqp = a.dq + (rowbase + qi) * a.lddq + head * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(dqt[4 * g] * a.scale, dqt[4 * g + 1] * a.scale, dqt[4 * g + 2] * a.scale,
                                   dqt[4 * g + 3] * a.scale);
            *reinterpret_cast<float4*>(dqp + 8 * g + 4 * hf) = t;
        }
    }
}

bool aligned16(const void* p, int64_t ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

}  // namespace

extern "C" int gaot_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int64_t ldq,
                             int64_t ldk, int64_t ldv, int64_t ldo, int B, int S, int H, int HKV, int head_dim,
                             float scale, int precision, gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_fwd: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(q && k && v && o && lse, "null pointer");
    GAOT_CHECK_ARG(aligned16(q, ldq) && aligned16(k, ldk) && aligned16(v, ldv) && aligned16(o, ldo),
                   "q/k/v/o must be 16-byte aligned with row strides that are multiples of 4 floats");
    AttnArgs a{q, k, v, o, lse, ldq, ldk, ldv, ldo, B, S, H, HKV, scale};
    dim3 grid((unsigned)ceil_div(S, 128), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(k_attn_fwd_f32, grid, dim3(256), 0, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                             const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldq, int64_t ldk,
                             int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int B,
                             int S, int H, int HKV, int head_dim, float scale, int precision, int phase_mask,
                             gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_bwd: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "null pointer");
    GAOT_CHECK_ARG(aligned16(q, ldq) && aligned16(k, ldk) && aligned16(v, ldv) && aligned16(o, ldo) &&
                       aligned16(d_o, lddo) && aligned16(dq, lddq) && aligned16(dk, lddk) && aligned16(dv, lddv),
                   "tensors must be 16-byte aligned with row strides that are multiples of 4 floats");
    AttnBwdArgs a{q, k, v, o, d_o, lse, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, B, S, H, HKV, scale};
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * S * H;
    if (phase_mask & 1) hipLaunchKernelGGL(k_attn_delta, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, a);
    if (phase_mask & 2)
        hipLaunchKernelGGL(k_attn_bwd_dkv_f32, dim3((unsigned)ceil_div(S, 128), (unsigned)HKV, (unsigned)B), dim3(256), 0, st, a);
    if (phase_mask & 4)
        hipLaunchKernelGGL(k_attn_bwd_dq_f32, dim3((unsigned)ceil_div(S, 128), (unsigned)H, (unsigned)B), dim3(256), 0, st, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
