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
#include "attn_dropout.h"

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
    gdrop::Drop drop;
};

struct AttnBwdArgs {
    const float* q; const float* k; const float* v; const float* o; const float* d_o; const float* lse;
    float* delta;                                // [B][H][S]
    float* dq; float* dk; float* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, S, H, HKV;
    float scale;
    gdrop::Drop drop;
};

// dropout words (csrc/attn_dropout.h).  Lanes that hold ONE query and the 32 keys of a tile in runs of 4 (forward,
// dQ) read their 8 key-pair words from bw_s, stored [hf][g][pair]; lanes that hold ONE key and runs of queries
// (dK/dV) read the row words of the tile's queries, split into halfword copies [parity][32].
__device__ __forceinline__ void stage_col_words(uint32_t* bw_s, uint32_t ck, int64_t k0) {
    if (threadIdx.x < 16) {
        const int jj = threadIdx.x;
        bw_s[((jj >> 1) & 1) * 8 + (jj >> 2) * 2 + (jj & 1)] = gdrop::col_word(ck, (uint32_t)(k0 >> 1) + jj);
    }
}
// keep flags of the lane's 16 rows (bit r = row r of the accumulator layout)
__device__ __forceinline__ uint32_t keep_bits_cols(uint32_t aw, const uint32_t* bw_s, int hf, uint32_t thr) {
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = aw ^ bw_s[hf * 8 + j];
        bits |= ((x & 0xffffu) >= thr ? 1u : 0u) << (2 * j);
        bits |= ((x >> 16) >= thr ? 1u : 0u) << (2 * j + 1);
    }
    return bits;
}

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
template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_fwd_f32(AttnArgs a) {
    __shared__ float Ks[32 * LDP];
    __shared__ float Vs[32 * LDP];
    __shared__ uint32_t bw_s[16];
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
    uint32_t aw = 0, ck = 0;
    if constexpr (DROP) {
        const unsigned long long seed = *a.drop.seed;
        const int bh = a.drop.bh(b, head);
        aw = gdrop::row_word(gdrop::row_key(seed, bh), (uint32_t)(q0 + l31));
        ck = gdrop::col_key(seed, bh);
    }

    float4 kt = tile_ld(kp, a.ldk, 0, a.S), vt = tile_ld(vp, a.ldv, 0, a.S);
    for (int64_t k0 = 0; k0 < a.S; k0 += 32) {
        __syncthreads();
        tile_st(Ks, LDP, kt);
        tile_st(Vs, LDP, vt);
        if constexpr (DROP) stage_col_words(bw_s, ck, k0);
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
        l = l * alpha + ps;   // the normaliser is the UNdropped row sum
        m = mn;
        if constexpr (DROP) {
            const uint32_t kb = keep_bits_cols(aw, bw_s, hf, a.drop.thr);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = ((kb >> r) & 1u) ? s[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= alpha;
        // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[mfma32_row(i, hf) * LDP + l31], s[i], acc, 0, 0, 0);
    }
    const int64_t qi = q0 + l31;
    if (qi < a.S) {
        const float inv = DROP ? a.drop.inv_keep / l : 1.f / l;
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
template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_f32(AttnBwdArgs a) {
    __shared__ float Qs[32 * LDP];
    __shared__ float dOs[32 * LDP];
    __shared__ float lse_s[32];
    __shared__ float del_s[32];
    __shared__ uint32_t aw_s[64];
    // dropout: dP = keep * (dO.V) / (1-p); the kernel forms (1-p) * dS (delta staged times (1-p)) and rescales
    // dK -- and dV, accumulated from keep * P -- by 1/(1-p) at the end
    const float dscale = DROP ? a.drop.keep : 1.f;
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
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
            et = (threadIdx.x < a.S) ? delp[threadIdx.x] * dscale : 0.f;
        }
        uint32_t rk = 0, bsel = 0;
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t bw = gdrop::col_word(gdrop::col_key(seed, bh), (uint32_t)(ki >> 1));
            bsel = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
        }
        for (int64_t q0 = 0; q0 < a.S; q0 += 32) {
            __syncthreads();
            tile_st(Qs, LDP, qt);
            tile_st(dOs, LDP, dt);
            if (threadIdx.x < 32) { lse_s[threadIdx.x] = lt; del_s[threadIdx.x] = et; }
            if constexpr (DROP) {
                if (threadIdx.x < 32) {
                    const uint32_t w = gdrop::row_word(rk, (uint32_t)q0 + threadIdx.x);
                    aw_s[threadIdx.x] = w & 0xffffu;
                    aw_s[32 + threadIdx.x] = w >> 16;
                }
            }
            __syncthreads();
            if (q0 + 32 < a.S) {
                qt = tile_ld(qp, a.ldq, q0 + 32, a.S);
                dt = tile_ld(dop, a.lddo, q0 + 32, a.S);
                if (threadIdx.x < 32) {
                    const int64_t qq = q0 + 32 + threadIdx.x;
                    lt = (qq < a.S) ? lsep[qq] * LOG2E : INFINITY;
                    et = (qq < a.S) ? delp[qq] * dscale : 0.f;
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
                bool keep = true;
                if constexpr (DROP) keep = (aw_s[(l31 & 1) * 32 + qr] ^ bsel) >= a.drop.thr;
                s[r] = keep ? p : 0.f;                                  // (kept) P
                dp[r] = p * ((keep ? dp[r] : 0.f) - del_s[qr]);         // dS (without the 1/sqrt(d) factor)
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
            const float vsc = DROP ? a.drop.inv_keep : 1.f, ksc = a.scale * vsc;
            float4 t = make_float4(dkt[4 * g] * ksc, dkt[4 * g + 1] * ksc, dkt[4 * g + 2] * ksc, dkt[4 * g + 3] * ksc);
            *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = t;
            float4 u = make_float4(dvt[4 * g] * vsc, dvt[4 * g + 1] * vsc, dvt[4 * g + 2] * vsc, dvt[4 * g + 3] * vsc);
            *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: block = 4 waves x 32 queries; grid (ceil(S/128), H, B)
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq_f32(AttnBwdArgs a) {
    __shared__ float Ks[32 * LDP];
    __shared__ float Vs[32 * LDP];
    __shared__ uint32_t bw_s[16];
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
    const float del = (qi < a.S) ? a.delta[((int64_t)b * a.H + head) * a.S + qi] * (DROP ? a.drop.keep : 1.f) : 0.f;
    f32x16 dqt;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
    uint32_t aw = 0, ck = 0;
    if constexpr (DROP) {
        const unsigned long long seed = *a.drop.seed;
        const int bh = a.drop.bh(b, head);
        aw = gdrop::row_word(gdrop::row_key(seed, bh), (uint32_t)qi);
        ck = gdrop::col_key(seed, bh);
    }

    float4 kt = tile_ld(kp, a.ldk, 0, a.S), vt = tile_ld(vp, a.ldv, 0, a.S);
    for (int64_t k0 = 0; k0 < a.S; k0 += 32) {
        __syncthreads();
        tile_st(Ks, LDP, kt);
        tile_st(Vs, LDP, vt);
        if constexpr (DROP) stage_col_words(bw_s, ck, k0);
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
        uint32_t kbits = 0xffffu;
        if constexpr (DROP) kbits = keep_bits_cols(aw, bw_s, hf, a.drop.thr);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = exp2f(s[r] - lse2);
            if (k0 + mfma32_row(r, hf) >= a.S) p = 0.f;
            dp[r] = p * ((((kbits >> r) & 1u) ? dp[r] : 0.f) - del);
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
            const float qsc = DROP ? a.scale * a.drop.inv_keep : a.scale;
            float4 t = make_float4(dqt[4 * g] * qsc, dqt[4 * g + 1] * qsc, dqt[4 * g + 2] * qsc, dqt[4 * g + 3] * qsc);
            *reinterpret_cast<float4*>(dqp + 8 * g + 4 * hf) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV AND dQ in one pass (round 6): the dK / dV kernel above already holds S, P, dP and dS of every (query tile, key block)
// pair -- the separate dQ pass recomputes all of them (3 of its 3 products; 7 S^2 d products per layer where 5 suffice).  Here a
// wave also forms dQ^T[d][q] += K^T[d][key] dS^T[key][q] for ITS 32 keys: dS goes through a wave-private LDS tile to get the keys
// onto the reduction axis, K of the block sits in LDS unscaled.  The waves' partials of a query tile are summed in wave order through
// the same LDS tiles (between the two barriers of the next tile's staging) and leave as one fp32 slab partial per workgroup:
// part[b][head][slab][S][32]; k_attn_dq_reduce_f32 sums the slabs in slab order and applies scale / (1-p).  No atomics, fixed
// order: bit-reproducible.  block = 8 waves x 32 keys; grid (ceil(S/256), HKV, B); dynamic LDS 8.4 + 2 x 33.8 KB.
// ------------------------------------------------------------------------------------------------
constexpr int FW = 8;                                     // waves (key blocks) per workgroup
constexpr int FUSED_LDS = (2 * 32 * LDP + 64 + 64) * 4 + 2 * FW * 32 * LDP * 4;
template <bool DROP>
__global__ __launch_bounds__(64 * FW, 1) void k_attn_bwd_fused_f32(AttnBwdArgs a, float* __restrict__ part, int nslab) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* Qs = fsm;
    float* dOs = Qs + 32 * LDP;
    float* lse_s = dOs + 32 * LDP;                       // [32]
    float* del_s = lse_s + 32;                           // [32]
    uint32_t* aw_s = reinterpret_cast<uint32_t*>(del_s + 32);   // [64]
    float* Ksw = del_s + 32 + 64;                        // [FW][32 keys][LDP]: K of the waves' blocks, unscaled
    float* dSw = Ksw + FW * 32 * LDP;                    // [FW][32][LDP]: dS [q][key] of the wave, then its dQ partial [q][d]
    const float dscale = DROP ? a.drop.keep : 1.f;
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.y, b = blockIdx.z, slab = blockIdx.x;
    const int rep = a.H / a.HKV;
    const int64_t key0 = (int64_t)blockIdx.x * (32 * FW) + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const float sc = a.scale * LOG2E;
    const int64_t ki = key0 + l31;
    float kreg[16], vreg[16];
    float* myK = Ksw + wave * 32 * LDP;
    float* myS = dSw + wave * 32 * LDP;
    {
        const float* kp = a.k + (rowbase + ki) * a.ldk + hkv * D;
        const float* vp = a.v + (rowbase + ki) * a.ldv + hkv * D;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float kv = (ki < a.S) ? kp[2 * i + hf] : 0.f;
            myK[l31 * LDP + 2 * i + hf] = kv;
            kreg[i] = kv * sc;
            vreg[i] = (ki < a.S) ? vp[2 * i + hf] : 0.f;
        }
    }
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }
    // a staging thread: threads 0..255 carry the Q tile, 256..511 the dO tile (one float4 each)
    const int sr = (threadIdx.x & 255) >> 3, sc4 = threadIdx.x & 7;
    const bool is_q = threadIdx.x < 256;

    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const float* tp = is_q ? a.q + rowbase * a.ldq + head * D : a.d_o + rowbase * a.lddo + head * D;
        const int64_t tld = is_q ? a.ldq : a.lddo;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        float* slabp = part + (((int64_t)b * a.H + head) * nslab + slab) * a.S * D;
        auto ld_tile = [&](int64_t q0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + sr < a.S) t = *reinterpret_cast<const float4*>(tp + (q0 + sr) * tld + 4 * sc4);
            return t;
        };
        // the sum of the eight waves' dQ partials of the tile at q0 (wave order), one coalesced 128-byte row per query
        auto flush = [&](int64_t q0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = threadIdx.x + 512 * e, q = idx >> 5, d = idx & 31;
                float acc = dSw[q * LDP + d];
#pragma unroll
                for (int w = 1; w < FW; ++w) acc += dSw[w * 32 * LDP + q * LDP + d];
                if (q0 + q < a.S) slabp[(q0 + q) * D + d] = acc;
            }
        };
        float4 tt = ld_tile(0);
        float lt = 0.f, et = 0.f;
        if (threadIdx.x < 32) {
            lt = (threadIdx.x < a.S) ? lsep[threadIdx.x] * LOG2E : INFINITY;
            et = (threadIdx.x < a.S) ? delp[threadIdx.x] * dscale : 0.f;
        }
        uint32_t rk = 0, bsel = 0;
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t bw = gdrop::col_word(gdrop::col_key(seed, bh), (uint32_t)(ki >> 1));
            bsel = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
        }
        for (int64_t q0 = 0; q0 < a.S; q0 += 32) {
            __syncthreads();                       // everyone is done with the previous tile: its dQ partials are in dSw
            if (q0 > 0) flush(q0 - 32);
            {
                float* pdst = (is_q ? Qs : dOs) + sr * LDP + 4 * sc4;
                pdst[0] = tt.x; pdst[1] = tt.y; pdst[2] = tt.z; pdst[3] = tt.w;
            }
            if (threadIdx.x < 32) { lse_s[threadIdx.x] = lt; del_s[threadIdx.x] = et; }
            if constexpr (DROP) {
                if (threadIdx.x < 32) {
                    const uint32_t w = gdrop::row_word(rk, (uint32_t)q0 + threadIdx.x);
                    aw_s[threadIdx.x] = w & 0xffffu;
                    aw_s[32 + threadIdx.x] = w >> 16;
                }
            }
            __syncthreads();
            if (q0 + 32 < a.S) {
                tt = ld_tile(q0 + 32);
                if (threadIdx.x < 32) {
                    const int64_t qq = q0 + 32 + threadIdx.x;
                    lt = (qq < a.S) ? lsep[qq] * LOG2E : INFINITY;
                    et = (qq < a.S) ? delp[qq] * dscale : 0.f;
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
                bool keep = true;
                if constexpr (DROP) keep = (aw_s[(l31 & 1) * 32 + qr] ^ bsel) >= a.drop.thr;
                s[r] = keep ? p : 0.f;                                  // (kept) P
                dp[r] = p * ((keep ? dp[r] : 0.f) - del_s[qr]);         // dS (without the 1/sqrt(d) factor)
                myS[qr * LDP + l31] = dp[r];                            // [q][key]
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qr = mfma32_row(i, hf);
                dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[qr * LDP + l31], s[i], dvt, 0, 0, 0);
                dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qr * LDP + l31], dp[i], dkt, 0, 0, 0);
            }
            // dQ^T[d][q] = sum over this wave's keys of K^T[d][key] dS^T[key][q]: lane = query, register = d
            f32x16 dqt;
#pragma unroll
            for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dqt = __builtin_amdgcn_mfma_f32_32x32x2f32(myK[(2 * i + hf) * LDP + l31], myS[l31 * LDP + 2 * i + hf], dqt, 0, 0, 0);
            // the wave's reads of its dS tile are issued: the tile becomes its dQ partial [q][d]
#pragma unroll
            for (int r = 0; r < 16; ++r) myS[l31 * LDP + mfma32_row(r, hf)] = dqt[r];
        }
        __syncthreads();
        flush(((a.S - 1) / 32) * 32);
    }
    if (ki < a.S) {
        float* dkp = a.dk + (rowbase + ki) * a.lddk + hkv * D;
        float* dvp = a.dv + (rowbase + ki) * a.lddv + hkv * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float vsc = DROP ? a.drop.inv_keep : 1.f, ksc = a.scale * vsc;
            float4 t = make_float4(dkt[4 * g] * ksc, dkt[4 * g + 1] * ksc, dkt[4 * g + 2] * ksc, dkt[4 * g + 3] * ksc);
            *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = t;
            float4 u = make_float4(dvt[4 * g] * vsc, dvt[4 * g + 1] * vsc, dvt[4 * g + 2] * vsc, dvt[4 * g + 3] * vsc);
            *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
        }
    }
}

// dq[b, q, head, :] = scale / (1-p) * sum over the key slabs (slab order) of the fused fp32 kernel's partials
__global__ void k_attn_dq_reduce_f32(const float* __restrict__ part, int nslab, int B, int S, int H, int64_t lddq, float qsc,
                                     float* __restrict__ dq) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b, head, q, 4-column group)
    const int64_t n = (int64_t)B * H * S * 8;
    if (i >= n) return;
    const int c = (int)(i & 7);
    const int64_t q = (i >> 3) % S, bh = (i >> 3) / S;
    const int head = (int)(bh % H);
    const int64_t b = bh / H;
    const float* p = part + (bh * nslab * (int64_t)S + q) * D + 4 * c;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= nslab; s += 8) {        // read once: non-temporal, eight slabs requested before the first is added (slab order)
        f4v v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + (int64_t)(s + j) * S * D));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; s < nslab; ++s) acc += *reinterpret_cast<const f4v*>(p + (int64_t)s * S * D);
    *reinterpret_cast<f4v*>(dq + (b * S + q) * lddq + head * D + 4 * c) = acc * qsc;
}

// keep[b][h][q][k] of the dropout mask, one byte per element (checks and the oracle comparison only)
__global__ void k_dropout_mask(gdrop::Drop d, int H, int S, int64_t n, unsigned char* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long seed = *d.seed;
    const int k = (int)(i % S);
    const int q = (int)((i / S) % S);
    const int bh = (int)(i / ((int64_t)S * S));
    keep[i] = gdrop::keep_elem(gdrop::row_key(seed, bh), gdrop::col_key(seed, bh), (uint32_t)q, (uint32_t)k, d.thr) ? 1 : 0;
}

// out = state; state += stride  (one launch per dropout call: the word a forward uses and the advance of the stream)
__global__ void k_seed_next(unsigned long long* __restrict__ state, unsigned long long stride, unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long v = *state;
        *out = v;
        *state = v + stride;
    }
}

// out[i] = state + i * stride (i < n); state += n * stride: the words of the next n dropout calls in ONE launch
__global__ void k_seed_block(unsigned long long* __restrict__ state, unsigned long long stride, int n, unsigned long long* __restrict__ out) {
    const unsigned long long v = *state;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = v + (unsigned long long)i * stride;
    __syncthreads();
    if (threadIdx.x == 0) *state = v + (unsigned long long)n * stride;
}

bool aligned16(const void* p, int64_t ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

}  // namespace

extern "C" int gaot_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int64_t ldq,
                             int64_t ldk, int64_t ldv, int64_t ldo, int B, int S, int H, int HKV, int head_dim,
                             float scale, float dropout_p, const unsigned long long* dropout_seed, int head0, int heads_total,
                             int precision, gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_fwd: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(q && k && v && o && lse, "null pointer");
    GAOT_CHECK_ARG(aligned16(q, ldq) && aligned16(k, ldk) && aligned16(v, ldv) && aligned16(o, ldo),
                   "q/k/v/o must be 16-byte aligned with row strides that are multiples of 4 floats");
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seed), "dropout_p in [0,1) and a seed");
    GAOT_CHECK_ARG(heads_total == 0 || (head0 >= 0 && head0 + H <= heads_total), "head0 + H <= heads_total");
    AttnArgs a{q, k, v, o, lse, ldq, ldk, ldv, ldo, B, S, H, HKV, scale, gdrop::make_drop(dropout_seed, dropout_p, H, head0, heads_total)};
    dim3 grid((unsigned)ceil_div(S, 128), (unsigned)H, (unsigned)B);
    if (a.drop.thr) GAOT_KLAUNCH(k_attn_fwd_f32<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else GAOT_KLAUNCH(k_attn_fwd_f32<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                             const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldq, int64_t ldk,
                             int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int B,
                             int S, int H, int HKV, int head_dim, float scale, float dropout_p,
                             const unsigned long long* dropout_seed, int head0, int heads_total, int precision,
                             int phase_mask, gaot_stream_t stream) {
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
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seed), "dropout_p in [0,1) and a seed");
    AttnBwdArgs a{q, k, v, o, d_o, lse, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, B, S, H, HKV, scale,
                  gdrop::make_drop(dropout_seed, dropout_p, H, head0, heads_total)};
    GAOT_CHECK_ARG(heads_total == 0 || (head0 >= 0 && head0 + H <= heads_total), "head0 + H <= heads_total");
    const bool drop = a.drop.thr != 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * S * H;
    if (phase_mask & 1) GAOT_KLAUNCH(k_attn_delta, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, a);
    if (phase_mask & 2) {
        const dim3 g((unsigned)ceil_div(S, 128), (unsigned)HKV, (unsigned)B);
        if (drop) GAOT_KLAUNCH(k_attn_bwd_dkv_f32<true>, g, dim3(256), 0, st, a);
        else GAOT_KLAUNCH(k_attn_bwd_dkv_f32<false>, g, dim3(256), 0, st, a);
    }
    if (phase_mask & 4) {
        const dim3 g((unsigned)ceil_div(S, 128), (unsigned)H, (unsigned)B);
        if (drop) GAOT_KLAUNCH(k_attn_bwd_dq_f32<true>, g, dim3(256), 0, st, a);
        else GAOT_KLAUNCH(k_attn_bwd_dq_f32<false>, g, dim3(256), 0, st, a);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// fp32 mode, dK, dV and dQ from ONE pass (k_attn_bwd_fused_f32) + the slab reduction; delta as in gaot_attn_bwd (phase 1, run first
// by the caller or with run_delta != 0).  scratch: gaot_attn_bwd_fused_f32_scratch_bytes(B, S, H) bytes, 16-byte aligned.
extern "C" int64_t gaot_attn_bwd_fused_f32_scratch_bytes(int B, int S, int H) {
    if (B <= 0 || S <= 0 || H <= 0) return 0;
    return (int64_t)B * H * ceil_div(S, 32 * FW) * S * D * 4;
}
extern "C" int gaot_attn_bwd_fused_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                                       const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldq, int64_t ldk,
                                       int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int B,
                                       int S, int H, int HKV, int head_dim, float scale, float dropout_p,
                                       const unsigned long long* dropout_seed, int head0, int heads_total, int run_delta,
                                       void* scratch, size_t scratch_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_bwd_fused_f32: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv && scratch, "null pointer");
    GAOT_CHECK_ARG(aligned16(q, ldq) && aligned16(k, ldk) && aligned16(v, ldv) && aligned16(o, ldo) &&
                       aligned16(d_o, lddo) && aligned16(dq, lddq) && aligned16(dk, lddk) && aligned16(dv, lddv) && aligned16(scratch, 4),
                   "tensors must be 16-byte aligned with row strides that are multiples of 4 floats");
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seed), "dropout_p in [0,1) and a seed");
    GAOT_CHECK_ARG(heads_total == 0 || (head0 >= 0 && head0 + H <= heads_total), "head0 + H <= heads_total");
    GAOT_CHECK_ARG((int64_t)scratch_bytes >= gaot_attn_bwd_fused_f32_scratch_bytes(B, S, H), "scratch too small");
    AttnBwdArgs a{q, k, v, o, d_o, lse, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, B, S, H, HKV, scale,
                  gdrop::make_drop(dropout_seed, dropout_p, H, head0, heads_total)};
    const bool drop = a.drop.thr != 0;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)k_attn_bwd_fused_f32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS);
        hipError_t e2 = hipFuncSetAttribute((const void*)k_attn_bwd_fused_f32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            gaot_set_error("gaot_attn_bwd_fused_f32: cannot set dynamic LDS %d", FUSED_LDS);
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int64_t n = (int64_t)B * S * H;
    if (run_delta) GAOT_KLAUNCH(k_attn_delta, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, a);
    const int nslab = (int)ceil_div(S, 32 * FW);
    const dim3 g((unsigned)nslab, (unsigned)HKV, (unsigned)B);
    if (drop) GAOT_KLAUNCH(k_attn_bwd_fused_f32<true>, g, dim3(64 * FW), FUSED_LDS, st, a, (float*)scratch, nslab);
    else GAOT_KLAUNCH(k_attn_bwd_fused_f32<false>, g, dim3(64 * FW), FUSED_LDS, st, a, (float*)scratch, nslab);
    const float qsc = scale * (drop ? a.drop.inv_keep : 1.f);
    GAOT_KLAUNCH(k_attn_dq_reduce_f32, dim3((unsigned)ceil_div(n * 8, 256)), dim3(256), 0, st, (const float*)scratch, nslab, B, S, H, lddq, qsc, dq);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_dropout_seed_next(unsigned long long* state, unsigned long long stride, unsigned long long* out,
                                      gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(state && out, "null pointer");
    GAOT_KLAUNCH(k_seed_next, dim3(1), dim3(64), 0, (hipStream_t)stream, state, stride, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// the seed words of the next n dropout calls at once (ABI 11): out[i] = *state + i * stride, *state += n * stride -- the values n calls
// of gaot_dropout_seed_next would hand out (a Transformer of L blocks draws its L attention seeds with one launch instead of L)
extern "C" int gaot_dropout_seed_block(unsigned long long* state, unsigned long long stride, int n, unsigned long long* out, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(state && out && n > 0, "bad argument");
    GAOT_KLAUNCH(k_seed_block, dim3(1), dim3(64), 0, (hipStream_t)stream, state, stride, n, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_attn_dropout_mask(const unsigned long long* dropout_seed, float dropout_p, int B, int H, int S,
                                      unsigned char* keep, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(B > 0 && H > 0 && S > 0 && dropout_seed && keep, "bad arguments");
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p in [0,1)");
    const int64_t n = (int64_t)B * H * S * S;
    GAOT_KLAUNCH(k_dropout_mask, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       gdrop::make_drop(dropout_seed, dropout_p, H), H, S, n, keep);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
