// bf16 matrix-core attention (v_mfma_f32_32x32x16_bf16, fp32 softmax / accumulators) for the latent
// Transformer -- same operator as attn.hip (reference src/model/layers/attn.py:110-127), precision 1.
//
// With head_dim 32 the kernel is bound by the softmax VALU work, not by the MFMAs, so the design
// minimises everything that is not exp/max/sum:
//   * a prep pass turns the fused fp32 q|k|v projection into ONE bf16 image [rows][(h+2hkv)*32]
//     with RoPE applied and q pre-multiplied by (1/sqrt(d))*log2(e): the kernels use exp2 directly;
//   * K/V (fwd, dQ) or Q/dO (dK/dV) stream through LDS as row-major 32x32 bf16 tiles (64-B rows,
//     16-B chunks XOR-swizzled by (row>>2)&3): ds_read_b128 serves the operands that contract over
//     head_dim, ds_read_b64_tr_b16 (hardware transpose) the operands that contract over rows --
//     one image, both uses, conflict-free;
//   * score tiles are computed transposed where needed so that the fp32 accumulator of one MFMA
//     chain, rounded to bf16 in registers, IS the B operand of the next chain (no LDS round trip
//     for P / dS).
// dQ has its own pass: no float atomics, bit-reproducible gradients.
#include <stdlib.h>

#include <type_traits>

#include "attn_dropout.h"
#include "common.h"
#include "tile32.h"

namespace {

constexpr int D = 32;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }
// max over the two lanes (l, l+32) that share an MFMA column, without going through LDS
__device__ __forceinline__ float max_halves(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

// cooperative stage of NT tiles (each [32][32] bf16) of a row-major bf16 matrix with row pitch ld (elements)
// tiles 0,1 come from matrix A (rows row0.., row0+32..), tiles 2,3 from matrix B (same rows)
__device__ __forceinline__ void stage_load4(uint4 (&regs)[2], const bf16_t* pa, int64_t lda, const bf16_t* pb, int64_t ldb,
                                            int64_t row0, int64_t nrows) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int t = idx >> 7, r = (idx & 127) >> 2, c = idx & 3;
        const int64_t row = row0 + 32 * (t & 1) + r;
        const bf16_t* base = (t < 2) ? pa : pb;
        const int64_t ld = (t < 2) ? lda : ldb;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < nrows) val = *reinterpret_cast<const uint4*>(base + row * ld + 8 * c);
        regs[it] = val;
    }
}
// TPM tiles per matrix (A tiles first, then B tiles); 128 16-B chunks per tile
template <int TPM>
__device__ __forceinline__ void stage_loadN(uint4 (&regs)[TPM], const bf16_t* pa, int64_t lda, const bf16_t* pb, int64_t ldb,
                                            int64_t row0, int64_t nrows) {
#pragma unroll
    for (int it = 0; it < TPM; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int t = idx >> 7, r = (idx & 127) >> 2, c = idx & 3;
        const int64_t row = row0 + 32 * (t % TPM) + r;
        const bf16_t* base = (t < TPM) ? pa : pb;
        const int64_t ld = (t < TPM) ? lda : ldb;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < nrows) val = *reinterpret_cast<const uint4*>(base + row * ld + 8 * c);
        regs[it] = val;
    }
}
template <int TPM>
__device__ __forceinline__ void stage_storeN(const uint4 (&regs)[TPM], char* lds) {
#pragma unroll
    for (int it = 0; it < TPM; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int t = idx >> 7, r = (idx & 127) >> 2, c = idx & 3;
        *reinterpret_cast<uint4*>(lds + t * TILE_BYTES + tile_off(r, c)) = regs[it];
    }
}
template <int NT>
__device__ __forceinline__ void stage_store(const uint4 (&regs)[(NT * 128 + 255) / 256], char* lds) {
#pragma unroll
    for (int it = 0; it < (NT * 128 + 255) / 256; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int t = idx >> 7, r = (idx & 127) >> 2, c = idx & 3;
        if (t < NT) *reinterpret_cast<uint4*>(lds + t * TILE_BYTES + tile_off(r, c)) = regs[it];
    }
}

// ------------------------------------------------------------------------------------------------
// prep: fp32 fused projection [rows][ld] -> bf16 image, RoPE on q,k heads (optional), q *= qscale
// ------------------------------------------------------------------------------------------------
__global__ void k_prep_qkv(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t rows, int ld, int nq_heads,
                           int nk_heads, int S, const float* __restrict__ freqs, float qscale) {
    const int pairs = ld / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * pairs) return;
    const int p = (int)(i % pairs);
    const int64_t row = i / pairs;
    const int col = 2 * p;
    const int head = col / D;
    float2 v = *reinterpret_cast<const float2*>(x + row * ld + col);
    if (freqs && head < nq_heads + nk_heads) {
        const float ang = (float)(row % S) * freqs[(col % D) >> 1];
        float sn, cs;
        sincosf(ang, &sn, &cs);
        v = make_float2(v.x * cs - v.y * sn, v.y * cs + v.x * sn);
    }
    if (head < nq_heads) { v.x *= qscale; v.y *= qscale; }
    const unsigned packed = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    *reinterpret_cast<unsigned*>(y + row * ld + col) = packed;
}

// prep for backward: dO fp32 -> bf16, delta[b][h][s] = sum_d dO*O (fp32)
__global__ void k_prep_do(const float* __restrict__ d_o, const float* __restrict__ o, bf16_t* __restrict__ dob,
                          float* __restrict__ delta, int B, int S, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, head)
    const int64_t n = (int64_t)B * S * H;
    if (i >= n) return;
    const int head = (int)(i % H);
    const int64_t row = i / H;
    const float* dp = d_o + row * H * D + head * D;
    const float* op = o + row * H * D + head * D;
    bf16_t* yp = dob + row * H * D + head * D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(dp + 4 * c);
        const float4 b = *reinterpret_cast<const float4*>(op + 4 * c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        const uint2 pk = make_uint2((unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16),
                                    (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16));
        *reinterpret_cast<uint2*>(yp + 4 * c) = pk;
    }
    delta[((row / S) * H + head) * S + (row % S)] = s;
}

// the same when dO already IS the bf16 image (sequence-parallel step: the gradient arrives as bf16 from the all-to-all):
// only delta[b][h][s] = sum_d dO * O is left to do
__global__ void k_delta_bf16(const bf16_t* __restrict__ dob, const float* __restrict__ o, float* __restrict__ delta, int B,
                             int S, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, head)
    const int64_t n = (int64_t)B * S * H;
    if (i >= n) return;
    const int head = (int)(i % H);
    const int64_t row = i / H;
    const bf16_t* dp = dob + row * H * D + head * D;
    const float* op = o + row * H * D + head * D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const uint4 a = *reinterpret_cast<const uint4*>(dp + 8 * c);
        const float4 b0 = *reinterpret_cast<const float4*>(op + 8 * c), b1 = *reinterpret_cast<const float4*>(op + 8 * c + 4);
        s += __uint_as_float(a.x << 16) * b0.x + __uint_as_float(a.x & 0xffff0000u) * b0.y + __uint_as_float(a.y << 16) * b0.z +
             __uint_as_float(a.y & 0xffff0000u) * b0.w + __uint_as_float(a.z << 16) * b1.x + __uint_as_float(a.z & 0xffff0000u) * b1.y +
             __uint_as_float(a.w << 16) * b1.z + __uint_as_float(a.w & 0xffff0000u) * b1.w;
    }
    delta[((row / S) * H + head) * S + (row % S)] = s;
}

struct FwdArgs {
    const bf16_t* qkv;   // bf16 image [B*S][ld]
    float* o;            // [B*S][H*32] fp32
    float* lse;          // [B][H][S]
    int64_t ld;
    int B, S, H, HKV;
    gdrop::Drop drop;
    // key-range split (few heads: one head is only S/128 workgroups): blockIdx.y = part, keys [part*chunk, +chunk);
    // part p writes its own normalised O / lse at o + p*o_part, lse + p*lse_part (combined by k_attn_combine)
    int chunk;
    int64_t o_part, lse_part;
    const float* kmax2;  // [B][KN_BLOCKS][HKV] partial maxima over the keys of |k|^2 (k_key_norm_max): bound of the scores
    int* redo;           // one flag per workgroup of the grid, written by the bound-based kernel: 1 = rows exceed the bound
};

// max_s |k_s|^2 per (batch, kv head) of the bf16 image, as KN_BLOCKS partial maxima per batch element (the forward kernel
// takes their maximum): out[b][block][hkv], one workgroup per (block, batch element, head).  A maximum does not depend on
// the order it is formed in: bit-reproducible without atomics, nothing to clear beforehand.
constexpr int KN_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_key_norm_max(const bf16_t* __restrict__ img, int64_t ld, int S, int H, int HKV,
                                                      float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.y, hkv = blockIdx.z;
    float m = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < S; row += (int64_t)KN_BLOCKS * 256) {
        const bf16_t* kp = img + ((int64_t)b * S + row) * ld + (H + hkv) * D;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            const uint4 a = *reinterpret_cast<const uint4*>(kp + 8 * c);
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
                n2 = fmaf(lo, lo, fmaf(hi, hi, n2));
            }
        }
        m = fmaxf(m, n2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[((int64_t)b * KN_BLOCKS + blockIdx.x) * HKV + hkv] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// dropout words of one 32-key tile for the lanes that hold ONE query and runs of 4 consecutive keys (rows
// 8g + 4hf + 0..3): 16 key-pair words per tile, stored as [hf][g][pair] so that a lane reads its 8 words with two
// ds_read_b128.  Called by the first 16*NTILES threads while they stage the keys k0...
template <int NTILES>
__device__ __forceinline__ void stage_col_words(uint32_t* bw_s, uint32_t ck, int64_t k0) {
    if (threadIdx.x < 16 * NTILES) {
        const int t = threadIdx.x >> 4, jj = threadIdx.x & 15;
        bw_s[t * 16 + ((jj >> 1) & 1) * 8 + (jj >> 2) * 2 + (jj & 1)] = gdrop::col_word(ck, (uint32_t)(k0 >> 1) + threadIdx.x);
    }
}
// v[r] = keep(q, key of row r) ? v[r] : other[r]   (aw = the lane's row word, wd = its 8 pair words of the tile)
template <bool ZERO>
__device__ __forceinline__ void drop_select(f32x16& v, const f32x16& other, uint32_t aw, const uint32_t* bw_tile, int hf,
                                            uint32_t thr) {
    const uint4 w0 = *reinterpret_cast<const uint4*>(bw_tile + hf * 8);
    const uint4 w1 = *reinterpret_cast<const uint4*>(bw_tile + hf * 8 + 4);
    const uint32_t wd[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = aw ^ wd[j];
        v[2 * j] = ((x & 0xffffu) >= thr) ? v[2 * j] : (ZERO ? 0.f : other[2 * j]);
        v[2 * j + 1] = ((x >> 16) >= thr) ? v[2 * j + 1] : (ZERO ? 0.f : other[2 * j + 1]);
    }
}

// zero the dropped elements of a P tile that is already packed to bf16 (dword j = rows 2j, 2j+1 = the two keys of
// pair word j): per dword one xor, one packed saturating 16-bit subtract (sign = KEPT), one packed arithmetic
// shift (sign -> 0xFFFF) and one and -- 2 VALU ops per element instead of compare + select on fp32.  The subtraction is
// (thr - 1) - x so that the plain two-operand v_and_b32 applies the mask (4 issue cycles; the and-not form is a three-operand
// encoding: 5 -- tools/lab/inst_cost.hip).
// awf = row word ^ 0x80008000 (halfwords as signed), tpk = (thr - 1 - 32768) in both halves (thr >= 1 whenever DROP runs).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void drop_packed(bf16x8& f0, bf16x8& f1, uint32_t awf, const uint32_t* bw_tile, int hf, s16x2 tpk) {
    const uint4 w0 = *reinterpret_cast<const uint4*>(bw_tile + hf * 8);
    const uint4 w1 = *reinterpret_cast<const uint4*>(bw_tile + hf * 8 + 4);
    const uint32_t wd[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    uint4 a = __builtin_bit_cast(uint4, f0), b = __builtin_bit_cast(uint4, f1);
    uint32_t pk[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const s16x2 d = __builtin_elementwise_sub_sat(tpk, __builtin_bit_cast(s16x2, awf ^ wd[j]));   // < 0 iff x > thr - 1 iff kept
        const s16x2 m = d >> (short)15;
        pk[j] &= __builtin_bit_cast(uint32_t, m);
    }
    f0 = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    f1 = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
}

// ------------------------------------------------------------------------------------------------
// forward: 4 waves x 32 queries; K/V in 64-key stages (2 tiles each)
// ------------------------------------------------------------------------------------------------
// The running reference value m of a row need not be its exact maximum: p = exp2(S - m) only has to stay in range.  A tile
// first takes the optimistic path -- exp and row sums with NO maximum formed; m moves only when a lane's 16 values sum to more
// than RESCALE_SUM = 2^6 (so no p above 64 is ever used; an overflowed exp shows up as an infinite sum), and then the
// scores are recomputed (two MFMAs) and everything at the old scale is rescaled once.  With "rescale whenever a row maximum
// grows" ~40 % of the tiles of a wave went through the rescale path at S = 16 384 on random scores; now a handful do.
// o = acc / l and lse = m ln2 + log l do not depend on where m sits.
constexpr float RESCALE_SUM = 64.0f;
// FSB: schedule pins of the bound-based (FAST) tile, bit i = __builtin_amdgcn_sched_barrier(0) at 1: the start of a tile,
// 2: after the exp stream (before packing / row sums / mask), 4: before the P V products
template <int OCC, int TPM, bool DROP, bool FAST, int FSB = 0>
__global__ __launch_bounds__(256, OCC) void k_attn_fwd_bf16(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TPM * TILE_BYTES];  // K tiles, then V tiles
    __shared__ __attribute__((aligned(16))) uint32_t bw_s[DROP ? 16 * TPM : 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    // head-fastest workgroup order: consecutive workgroup ids go to consecutive XCDs, so with 8 heads every XCD
    // streams ONE head's K / V (2 MB at S = 16384) through its own 4-MB L2 instead of all eight heads' 16 MB
    const int head = blockIdx.x % a.H, b = blockIdx.z;
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)(blockIdx.x / a.H) * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
    const bf16_t* kp = a.qkv + rowbase * a.ld + (a.H + hkv) * D;
    const bf16_t* vp = a.qkv + rowbase * a.ld + (a.H + a.HKV + hkv) * D;

    bf16x8 qf[2];
    {
        const int64_t qi = q0 + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (qi < a.S) qf[s] = *reinterpret_cast<const bf16x8*>(qp + qi * a.ld + 16 * s + 8 * hf);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[s][j] = 0;
            }
        }
    }
    f32x16 acc, negm;   // negm = -m broadcast: C operand of the score MFMA, so the tile arrives as S - m
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; negm[r] = 0.f; }
    float m = 0.f, l = 0.f;
    uint32_t aw = 0, ck = 0;
    if constexpr (DROP) {
        const unsigned long long seed = *a.drop.seed;
        const int bh = a.drop.bh(b, head);
        aw = gdrop::row_word(gdrop::row_key(seed, bh), (uint32_t)(q0 + l31)) ^ 0x80008000u;
        ck = gdrop::col_key(seed, bh);
    }
    const short ts = (short)((int)a.drop.thr - 1 - 32768);
    const s16x2 tpk = {ts, ts};

    // Static bound (FAST, the common case): the image's q rows carry scale * log2(e), so a score is at most |q| max_s |k_s| in
    // log2 units (Cauchy-Schwarz on the very bf16 values the MFMA multiplies).  While that bound stays below 54 for every row
    // of the workgroup, p = exp2(S) needs NO reference value at all: every p, the row sums (< S 2^54) and the O accumulator stay
    // far inside fp32, and o = acc / l, lse = log l do not depend on a common factor.  The tile then costs 16 exp, 8 packs and
    // the mask and 16 adds for the row sums -- no maximum, no subtraction, no overflow test; without the adaptive path's state the
    // kernel fits 128 registers, so all 1024 workgroups of the shipped shape are resident at once (4 per CU, no second round).
    // A workgroup whose bound is larger raises its flag and leaves; the adaptive kernel (!FAST, launched right after on the
    // same grid) does exactly the flagged workgroups.
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if constexpr (FAST) {
        constexpr float BOUND2 = 54.f * 54.f;
        float q2 = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = __uint_as_float((unsigned)(unsigned short)qf[s][j] << 16);
                q2 = fmaf(v, v, q2);
            }
        q2 += xhalf(q2);
        float k2 = a.kmax2[((int64_t)b * KN_BLOCKS + lane) * a.HKV + hkv];   // KN_BLOCKS = 64 partial maxima, one per lane
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k2 = fmaxf(k2, __shfl_xor(k2, o, 64));
        const int over = __syncthreads_or(!(q2 * k2 <= BOUND2));
        if (threadIdx.x == 0) a.redo[wg] = over ? 1 : 0;
        if (over) return;
    } else {
        if (a.redo[wg] == 0) return;
    }

    // one 32-key tile.  TAIL: keys >= S are masked.  Fast path (no running max grows): p = exp2(S - m) needs
    // no subtraction and the O accumulator is not rescaled; otherwise the max moves and everything at the old
    // scale (acc, l) is rescaled exactly once.
    const int64_t lo = (int64_t)blockIdx.y * a.chunk, hi = min((int64_t)a.S, lo + a.chunk);
    uint4 regs[TPM];
    stage_loadN<TPM>(regs, kp, a.ld, vp, a.ld, lo, a.S);
    for (int64_t k0 = lo; k0 < hi; k0 += 32 * TPM) {
        __syncthreads();
        stage_storeN<TPM>(regs, lds);
        if constexpr (DROP) stage_col_words<TPM>(bw_s, ck, k0);
        __syncthreads();
        if (k0 + 32 * TPM < hi) stage_loadN<TPM>(regs, kp, a.ld, vp, a.ld, k0 + 32 * TPM, a.S);
        // one 32-key tile at a time.  Fast path (no running max grows): p = exp2(S - m) needs no subtraction and
        // the O accumulator is not rescaled; otherwise everything at the old scale (acc, l, this tile) is rescaled
        // exactly once, in place.
        auto do_tile = [&](int t, int64_t kb) {
            const char* kt = lds + t * TILE_BYTES;
            const char* vt = lds + (TPM + t) * TILE_BYTES;
            const bool first = (kb == lo);
            f32x16 sc;
            auto scores = [&]() {   // S - m for this tile (keys past the end of the sequence masked)
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 0), qf[0], negm, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 1), qf[1], sc, 0, 0, 0);
                if (kb + 32 > a.S) {   // wave-uniform: only the last tile of the sequence
                    const int nv = (int)(a.S - kb);
    #pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (mfma32_row(r, hf) >= nv) sc[r] = -INFINITY;
                }
            };
            if constexpr (FAST) {
                if constexpr ((FSB & 1) != 0) __builtin_amdgcn_sched_barrier(0);
                f32x16 zero;
    #pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 0), qf[0], zero, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 1), qf[1], sc, 0, 0, 0);
                if (kb + 32 > a.S) {   // wave-uniform: only the last tile of the sequence
                    const int nv = (int)(a.S - kb);
    #pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (mfma32_row(r, hf) >= nv) sc[r] = -INFINITY;
                }
    #pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                if constexpr ((FSB & 2) != 0) __builtin_amdgcn_sched_barrier(0);
                bf16x8 p0, p1;
                acc_to_frags(sc, p0, p1);
                if constexpr ((FSB & 8) != 0) {   // row sums of the PACKED (bf16-rounded, undropped) p -- the values the P V product uses: one v_dot2c_f32_bf16 per pair
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
                    const uint4 ua = __builtin_bit_cast(uint4, p0), ub = __builtin_bit_cast(uint4, p1);
                    const uint32_t pw[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
                    float q0 = 0.f, q1 = 0.f;
    #pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        q0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pw[j]), ones, q0, false);
                        q1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pw[j + 1]), ones, q1, false);
                    }
                    l += q0 + q1;
                } else {   // l stays undropped; two chains (a ones-row MFMA with the packed P was measured: 2-4 % slower than these adds)
                    float q0 = 0.f, q1 = 0.f;
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) { q0 += sc[r]; q1 += sc[r + 1]; }
                    l += q0 + q1;
                }
                if constexpr (DROP) drop_packed(p0, p1, aw, bw_s + 16 * t, hf, tpk);
                if constexpr ((FSB & 4) != 0) __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(vt, lane, 0), p0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(vt, lane, 1), p1, acc, 0, 0, 0);
                return;
            }
            float ps0 = 0.f, ps1 = 0.f;   // two scalar chains: packed-f32 adds cost more issue cycles beside MFMAs
            auto exp_sum = [&]() {
                ps0 = 0.f; ps1 = 0.f;
    #pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                    sc[r + 1] = __builtin_amdgcn_exp2f(sc[r + 1]);
                    ps0 += sc[r];
                    ps1 += sc[r + 1];
                }
            };
            scores();
            bool slow = first;
            if (!first) {
                // optimistic path: no maximum is formed at all.  Every p is positive, so a lane whose 16 values sum to at most
                // 2^RESCALE_THR holds no p above it; a larger (or infinite) sum sends the wave through the path below, which
                // recomputes the scores (two MFMAs) and moves the reference values.
                exp_sum();
                if (__any(!(ps0 + ps1 <= RESCALE_SUM))) {
                    scores();
                    slow = true;
                }
            }
            if (slow) {
                float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
    #pragma unroll
                for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
                mx = fmaxf(mx, sc[15]);
                mx = max_halves(mx);
                // rescale everything that is at the old scale, in place
                const float up = first ? mx : fmaxf(mx, 0.f);     // how far this lane's reference value moves (log2 units)
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-up);
                m += up;
                l *= alpha;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[r] -= up;
                    acc[r] *= alpha;
                    negm[r] = -m;
                }
                exp_sum();
            }
            l += ps0 + ps1;   // per-half partial; the halves are added once, after the key loop
            bf16x8 p0, p1;
            acc_to_frags(sc, p0, p1);
            if constexpr (DROP) drop_packed(p0, p1, aw, bw_s + 16 * t, hf, tpk);   // l stays undropped
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(vt, lane, 0), p0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(vt, lane, 1), p1, acc, 0, 0, 0);
        };
        if constexpr (OCC >= 5) {   // rolled: fewer live registers, one more wave per SIMD
#pragma unroll 1
            for (int t = 0; t < TPM; ++t) {
                const int64_t kb = k0 + 32 * t;
                if (kb >= hi) break;
                do_tile(t, kb);
            }
        } else {
#pragma unroll
            for (int t = 0; t < TPM; ++t) {
                const int64_t kb = k0 + 32 * t;
                if (kb >= hi) break;
                do_tile(t, kb);
            }
        }
    }
    const int64_t qi = q0 + l31;
    l += xhalf(l);
    if (qi < a.S) {
        const float inv = DROP ? a.drop.inv_keep / l : 1.f / l;
        float* op = a.o + blockIdx.y * a.o_part + (rowbase + qi) * (a.H * D) + head * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hf) = t;
        }
        if (hf == 0) a.lse[blockIdx.y * a.lse_part + ((int64_t)b * a.H + head) * a.S + qi] = m * LN2 + logf(l);
    }
}

struct BwdArgs {
    const bf16_t* qkv;   // bf16 image (q pre-scaled by scale*log2e)
    const bf16_t* dob;   // bf16 dO [B*S][H*32]
    const float* lse;    // [B][H][S]
    const float* delta;  // [B][H][S]
    float* dqkv;         // fp32 [B*S][ld]
    int64_t ld;
    int B, S, H, HKV;
    float scale;
    gdrop::Drop drop;
    const float* freqs;   // RoPE frequencies or NULL: dq / dk leave the kernels rotated back (inverse RoPE in the epilogue)
    // range split of the streamed side (queries for dK/dV, keys for dQ) over blockIdx.y; part p accumulates into
    // dqkv + p*dqkv_part, the parts are summed in a fixed order by k_sum_parts
    int chunk;
    int64_t dqkv_part;
};

// inverse RoPE of the four consecutive head-dim columns 8g + 4hf .. + 3 (two rotation pairs) of the row at position
// pos: the gradient w.r.t. the UNrotated q / k (same arithmetic as k_rope(inverse) in rowops.hip)
__device__ __forceinline__ float4 unrope4(float4 t, const float* __restrict__ freqs, int pos, int g, int hf) {
    if (!freqs) return t;
    const int j = 4 * g + 2 * hf;
    float s0, c0, s1, c1;
    sincosf((float)pos * freqs[j], &s0, &c0);
    sincosf((float)pos * freqs[j + 1], &s1, &c1);
    return make_float4(t.x * c0 + t.y * s0, t.y * c0 - t.x * s0, t.z * c1 + t.w * s1, t.w * c1 - t.z * s1);
}

// dropout words for the lanes that hold ONE key and runs of queries (dK/dV): the row words of the staged queries,
// split into their two halfwords (copy 0 = low, copy 1 = high) so that a lane reads the copy its key's parity selects
template <int QS>
__device__ __forceinline__ void stage_row_words(uint32_t* aw_s, uint32_t rk, int64_t q0) {
    if (threadIdx.x < QS) {
        const uint32_t w = gdrop::row_word(rk, (uint32_t)q0 + threadIdx.x);
        aw_s[threadIdx.x] = w & 0xffffu;
        aw_s[QS + 4 + threadIdx.x] = w >> 16;   // + 4 words: the odd-key lanes' 16-byte reads hit other banks than the even ones'

    }
}
__device__ __forceinline__ void load_row_words(uint32_t (&w)[16], const uint32_t* rows32, int hf) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const uint4 v = *reinterpret_cast<const uint4*>(rows32 + 8 * g4 + 4 * hf);
        w[4 * g4] = v.x; w[4 * g4 + 1] = v.y; w[4 * g4 + 2] = v.z; w[4 * g4 + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV: 4 waves x 32 keys; Q / dO tiles stream through LDS; grid (ceil(S/128), HKV, B)
// ------------------------------------------------------------------------------------------------
template <int OCC, bool DROP>
__global__ __launch_bounds__(256, OCC) void k_attn_bwd_dkv_bf16(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // Q0 Q1 dO0 dO1
    __shared__ __attribute__((aligned(16))) float lse_s[64];
    __shared__ __attribute__((aligned(16))) float del_s[64];
    __shared__ __attribute__((aligned(16))) uint32_t aw_s[DROP ? 128 + 4 : 4];
    // with dropout the dP tile is (keep ? dO.V : 0) / (1-p) - delta; the kernel works with (1-p) times that, i.e.
    // delta is staged pre-multiplied by (1-p), and dK is rescaled by 1/(1-p) once at the end (dV likewise)
    const float dscale = DROP ? a.drop.keep : 1.f;
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.x % a.HKV, b = blockIdx.z;   // head-fastest order: one (kv) head per XCD (see k_attn_fwd_bf16)
    const int rep = a.H / a.HKV;
    const int64_t key0 = (int64_t)(blockIdx.x / a.HKV) * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const int64_t ki = key0 + l31;
    bf16x8 kf[2], vf[2];
    {
        const bf16_t* kp = a.qkv + (rowbase + ki) * a.ld + (a.H + hkv) * D;
        const bf16_t* vp = a.qkv + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (ki < a.S) {
                kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s + 8 * hf);
                vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s + 8 * hf);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kf[s][j] = 0; vf[s][j] = 0; }
            }
        }
    }
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }

    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
        const bf16_t* dop = a.dob + rowbase * (a.H * D) + head * D;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        uint4 regs[2];
        const int64_t lo = (int64_t)blockIdx.y * a.chunk, hi = min((int64_t)a.S, lo + a.chunk);
        stage_load4(regs, qp, a.ld, dop, (int64_t)a.H * D, lo, a.S);
        // row constants loaded RAW one stage ahead (clamped address, no arithmetic on the value: the wait then falls at the
        // store a stage later, not right behind the load -- see k_attn_bwd_fused); negated and scaled when staged
        float lt = 0.f, et = 0.f;
        auto load_consts = [&](int64_t qbase) {
            if (threadIdx.x < 64) {
                int64_t qq = qbase + threadIdx.x;
                qq = qq < a.S ? qq : (int64_t)a.S - 1;
                lt = lsep[qq];
                et = delp[qq];
            }
        };
        load_consts(lo);
        uint32_t rk = 0, bsel = 0;
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t bw = gdrop::col_word(gdrop::col_key(seed, bh), (uint32_t)(ki >> 1));
            bsel = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
        }
        for (int64_t q0 = lo; q0 < hi; q0 += 64) {
            __syncthreads();
            stage_store<4>(regs, lds);
            if (threadIdx.x < 64) {
                const bool in = q0 + threadIdx.x < a.S;
                lse_s[threadIdx.x] = in ? -lt * LOG2E : -INFINITY;   // staged NEGATED: they are the
                del_s[threadIdx.x] = in ? -et * dscale : 0.f;         // initial accumulator values
            }
            if constexpr (DROP) stage_row_words<64>(aw_s, rk, q0);
            __syncthreads();
            if (q0 + 64 < hi) {
                stage_load4(regs, qp, a.ld, dop, (int64_t)a.H * D, q0 + 64, a.S);
                load_consts(q0 + 64);
            }
#pragma unroll 1
            for (int t = 0; t < 2; ++t) {
                if (q0 + 32 * t >= hi) break;
                const char* qt = lds + t * TILE_BYTES;
                const char* dt = lds + (2 + t) * TILE_BYTES;
                // accumulators start at -lse[q] / -delta[q] (rows of this lane: 4 runs of 4 consecutive q)
                f32x16 sc, dp, dneg;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 lv = *reinterpret_cast<const float4*>(&lse_s[32 * t + 8 * g4 + 4 * hf]);
                    const float4 dv = *reinterpret_cast<const float4*>(&del_s[32 * t + 8 * g4 + 4 * hf]);
                    sc[4 * g4] = lv.x; sc[4 * g4 + 1] = lv.y; sc[4 * g4 + 2] = lv.z; sc[4 * g4 + 3] = lv.w;
                    dp[4 * g4] = dv.x; dp[4 * g4 + 1] = dv.y; dp[4 * g4 + 2] = dv.z; dp[4 * g4 + 3] = dv.w;
                }
                dneg = dp;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(qt, l31, hf, ks), kf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dt, l31, hf, ks), vf[ks], dp, 0, 0, 0);
                }
                if constexpr (DROP) {
                    const uint32_t* awp = aw_s + (l31 & 1) * (64 + 4) + 32 * t + 4 * hf;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint4 wv = *reinterpret_cast<const uint4*>(awp + 8 * g4);
                        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = 4 * g4 + i;
                            const float p = __builtin_amdgcn_exp2f(sc[r]);
                            const bool keep = (ww[i] ^ bsel) >= a.drop.thr;
                            float pm = keep ? p : 0.f;
                            asm volatile("" : "+v"(pm));   // select in fp32, then ONE cvt_pk per pair (see k_attn_bwd_dkv_kb)
                            sc[r] = pm;
                            dp[r] = p * (keep ? dp[r] : dneg[r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sc[r]);   // rows beyond S carry lse = +inf -> p = 0
                        sc[r] = p;
                        dp[r] = p * dp[r];
                    }
                }
                bf16x8 p0, p1, d0, d1;
                acc_to_frags(sc, p0, p1);
                acc_to_frags(dp, d0, d1);
                dvt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(dt, lane, 0), p0, dvt, 0, 0, 0);
                dvt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(dt, lane, 1), p1, dvt, 0, 0, 0);
                dkt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(qt, lane, 0), d0, dkt, 0, 0, 0);
                dkt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(qt, lane, 1), d1, dkt, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (ki < a.S) {
        float* dkp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + hkv) * D;
        float* dvp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
        const float vsc = DROP ? a.drop.inv_keep : 1.f;
        const float ksc = vsc / LOG2E;  // Q image carries scale*log2e: dK = dS^T (Q*scale) = dS^T Qimg / log2e
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(dkt[4 * g] * ksc, dkt[4 * g + 1] * ksc, dkt[4 * g + 2] * ksc, dkt[4 * g + 3] * ksc);
            *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)ki, g, hf);
            float4 u = make_float4(dvt[4 * g] * vsc, dvt[4 * g + 1] * vsc, dvt[4 * g + 2] * vsc, dvt[4 * g + 3] * vsc);
            *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
        }
    }
}

// row constants of one 32-row tile in accumulator layout (rows (r&3) + 8(r>>2) + 4hf): 4 runs of 4 rows
__device__ __forceinline__ void load_row_consts(f32x16& acc, const float* rows32, int hf) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const float4 v = *reinterpret_cast<const float4*>(rows32 + 8 * g4 + 4 * hf);
        acc[4 * g4] = v.x; acc[4 * g4 + 1] = v.y; acc[4 * g4 + 2] = v.z; acc[4 * g4 + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV with KB key blocks per wave (4 waves x KB x 32 keys): one LDS read of a Q / dO fragment or of the
// -lse / -delta row constants serves KB (query tile, key block) units, and the units of one query tile are
// independent instruction streams the scheduler can overlap (S/dP of block 1 under the exp work of block 0).
// Measured at S=16384, H=8: 0.59 ms vs 0.69 ms for one key block per wave (same box); removing the row
// constant and transposed-fragment reads from the one-block kernel altogether gave 0.58 ms, finer hand
// interleaving of the MFMA and exp streams nothing -- the kernel is bound by LDS traffic and its waits.
// ------------------------------------------------------------------------------------------------
template <int KB, int NT, int OCC, bool DROP>
__global__ __launch_bounds__(256, OCC) void k_attn_bwd_dkv_kb(BwdArgs a) {
    constexpr int QS = 32 * NT;
    __shared__ __attribute__((aligned(16))) char lds[2 * NT * TILE_BYTES];  // Q tiles, then dO tiles
    __shared__ __attribute__((aligned(16))) float lse_s[QS];
    __shared__ __attribute__((aligned(16))) float del_s[QS];
    __shared__ __attribute__((aligned(16))) uint32_t aw_s[DROP ? 2 * QS + 4 : 4];
    const float dscale = DROP ? a.drop.keep : 1.f;   // see k_attn_bwd_dkv_bf16
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.x % a.HKV, b = blockIdx.z;   // head-fastest order: one (kv) head per XCD
    const int rep = a.H / a.HKV;
    const int64_t key0 = (int64_t)(blockIdx.x / a.HKV) * (128 * KB) + wave * (32 * KB);
    const int64_t rowbase = (int64_t)b * a.S;
    bf16x8 kf[KB][2], vf[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int64_t ki = key0 + 32 * kb + l31;
        const bf16_t* kp = a.qkv + (rowbase + ki) * a.ld + (a.H + hkv) * D;
        const bf16_t* vp = a.qkv + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (ki < a.S) {
                kf[kb][s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s + 8 * hf);
                vf[kb][s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s + 8 * hf);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kf[kb][s][j] = 0; vf[kb][s][j] = 0; }
            }
        }
    }
    f32x16 dkt[KB], dvt[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[kb][r] = 0.f; dvt[kb][r] = 0.f; }

    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
        const bf16_t* dop = a.dob + rowbase * (a.H * D) + head * D;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        uint4 regs[NT];
        stage_loadN<NT>(regs, qp, a.ld, dop, (int64_t)a.H * D, 0, a.S);
        float lt = 0.f, et = 0.f;     // raw, one stage ahead (see k_attn_bwd_dkv_bf16)
        auto load_consts = [&](int64_t qbase) {
            if (threadIdx.x < QS) {
                int64_t qq = qbase + threadIdx.x;
                qq = qq < a.S ? qq : (int64_t)a.S - 1;
                lt = lsep[qq];
                et = delp[qq];
            }
        };
        load_consts(0);
        uint32_t rk = 0, bsel[KB];
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t ck = gdrop::col_key(seed, bh);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int64_t ki = key0 + 32 * kb + l31;
                const uint32_t bw = gdrop::col_word(ck, (uint32_t)(ki >> 1));
                bsel[kb] = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
            }
        }
        for (int64_t q0 = 0; q0 < a.S; q0 += QS) {
            __syncthreads();
            stage_storeN<NT>(regs, lds);
            if (threadIdx.x < QS) {
                const bool in = q0 + threadIdx.x < a.S;
                lse_s[threadIdx.x] = in ? -lt * LOG2E : -INFINITY;
                del_s[threadIdx.x] = in ? -et * dscale : 0.f;
            }
            if constexpr (DROP) stage_row_words<QS>(aw_s, rk, q0);
            __syncthreads();
            if (q0 + QS < a.S) {
                stage_loadN<NT>(regs, qp, a.ld, dop, (int64_t)a.H * D, q0 + QS, a.S);
                load_consts(q0 + QS);
            }
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                if (q0 + 32 * t >= a.S) break;
                const char* qt = lds + t * TILE_BYTES;
                const char* dt = lds + (NT + t) * TILE_BYTES;
                f32x16 lc, dc;
                load_row_consts(lc, lse_s + 32 * t, hf);
                load_row_consts(dc, del_s + 32 * t, hf);
                const bf16x8 qa0 = frag_rows(qt, l31, hf, 0), da0 = frag_rows(dt, l31, hf, 0);
                const bf16x8 qa1 = frag_rows(qt, l31, hf, 1), da1 = frag_rows(dt, l31, hf, 1);
                f32x16 sc[KB], dp[KB];
                // the eight score / dP MFMAs issue at raised priority: the other wave of the SIMD is then in its (longer) VALU
                // phase and fills the gaps (-1.3 %; the same around the dV / dK MFMAs, in dQ and in the forward: nothing or worse)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0, kf[kb][0], lc, 0, 0, 0);
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, vf[kb][0], dc, 0, 0, 0);
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1, kf[kb][1], sc[kb], 0, 0, 0);
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, vf[kb][1], dp[kb], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                const bf16x8 dc0 = frag_cols(dt, lane, 0), dc1 = frag_cols(dt, lane, 1);
                const bf16x8 qc0 = frag_cols(qt, lane, 0), qc1 = frag_cols(qt, lane, 1);
                if constexpr (DROP) {   // row words read 4 at a time: the two key blocks share them
                    const uint32_t* awp = aw_s + (l31 & 1) * (QS + 4) + 32 * t + 4 * hf;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint4 wv = *reinterpret_cast<const uint4*>(awp + 8 * g4);
                        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r = 4 * g4 + i;
                                const float p = __builtin_amdgcn_exp2f(sc[kb][r]);
                                const bool keep = (ww[i] ^ bsel[kb]) >= a.drop.thr;
                                float pm = keep ? p : 0.f;
                                // keep the select on the fp32 value: without this the compiler rounds every p to bf16 on its own,
                                // selects on the halves and re-packs them (2 cvt + 1 perm per pair instead of 1 cvt_pk)
                                asm volatile("" : "+v"(pm));
                                sc[kb][r] = pm;
                                dp[kb][r] = p * (keep ? dp[kb][r] : dc[r]);
                            }
                    }
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    if constexpr (!DROP) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float p = __builtin_amdgcn_exp2f(sc[kb][r]);   // rows beyond S carry lse = +inf -> p = 0
                            sc[kb][r] = p;
                            dp[kb][r] = p * dp[kb][r];
                        }
                    }
                    bf16x8 p0, p1, d0, d1;
                    acc_to_frags(sc[kb], p0, p1);
                    acc_to_frags(dp[kb], d0, d1);
                    dvt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dc0, p0, dvt[kb], 0, 0, 0);
                    dkt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qc0, d0, dkt[kb], 0, 0, 0);
                    dvt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dc1, p1, dvt[kb], 0, 0, 0);
                    dkt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qc1, d1, dkt[kb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const float vsc = DROP ? a.drop.inv_keep : 1.f;
    const float ksc = vsc / LOG2E;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int64_t ki = key0 + 32 * kb + l31;
        if (ki < a.S) {
            float* dkp = a.dqkv + (rowbase + ki) * a.ld + (a.H + hkv) * D;
            float* dvp = a.dqkv + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(dkt[kb][4 * g] * ksc, dkt[kb][4 * g + 1] * ksc, dkt[kb][4 * g + 2] * ksc, dkt[kb][4 * g + 3] * ksc);
                *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)ki, g, hf);
                float4 u = make_float4(dvt[kb][4 * g] * vsc, dvt[kb][4 * g + 1] * vsc, dvt[kb][4 * g + 2] * vsc, dvt[kb][4 * g + 3] * vsc);
                *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: 4 waves x 32 queries; K / V tiles stream through LDS; grid (ceil(S/128), H, B)
// ------------------------------------------------------------------------------------------------
template <int OCC, bool DROP>
__global__ __launch_bounds__(256, OCC) void k_attn_bwd_dq_bf16(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // K0 K1 V0 V1
    __shared__ __attribute__((aligned(16))) uint32_t bw_s[DROP ? 32 : 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    // head-fastest workgroup order: consecutive workgroup ids go to consecutive XCDs, so with 8 heads every XCD
    // streams ONE head's K / V (2 MB at S = 16384) through its own 4-MB L2 instead of all eight heads' 16 MB
    const int head = blockIdx.x % a.H, b = blockIdx.z;
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)(blockIdx.x / a.H) * 128 + wave * 32;
    const int64_t rowbase = (int64_t)b * a.S;
    const bf16_t* kp = a.qkv + rowbase * a.ld + (a.H + hkv) * D;
    const bf16_t* vp = a.qkv + rowbase * a.ld + (a.H + a.HKV + hkv) * D;
    const int64_t qi = q0 + l31;
    bf16x8 qf[2], dof[2];
    {
        const bf16_t* qp = a.qkv + (rowbase + qi) * a.ld + head * D;
        const bf16_t* dop = a.dob + (rowbase + qi) * (a.H * D) + head * D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (qi < a.S) {
                qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * hf);
                dof[s] = *reinterpret_cast<const bf16x8*>(dop + 16 * s + 8 * hf);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { qf[s][j] = 0; dof[s][j] = 0; }
            }
        }
    }
    const float lse2 = (qi < a.S) ? a.lse[((int64_t)b * a.H + head) * a.S + qi] * LOG2E : INFINITY;
    // dropout: the tile is (1-p) * dS, see k_attn_bwd_dkv_bf16; dQ is rescaled once at the end
    const float del = (qi < a.S) ? a.delta[((int64_t)b * a.H + head) * a.S + qi] * (DROP ? a.drop.keep : 1.f) : 0.f;
    // -lse[q] and -delta[q] live in two accumulator-shaped register sets and enter the MFMA chains as the C
    // operand: the score tile arrives as S - lse and the dP tile as dP - delta, at no VALU cost per tile
    f32x16 dqt, negl, negd;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dqt[r] = 0.f; negl[r] = -lse2; negd[r] = -del; }
    uint32_t aw = 0, ck = 0;
    if constexpr (DROP) {
        const unsigned long long seed = *a.drop.seed;
        const int bh = a.drop.bh(b, head);
        aw = gdrop::row_word(gdrop::row_key(seed, bh), (uint32_t)qi);
        ck = gdrop::col_key(seed, bh);
    }

    uint4 regs[2];
    const int64_t lo = (int64_t)blockIdx.y * a.chunk, hi = min((int64_t)a.S, lo + a.chunk);
    stage_load4(regs, kp, a.ld, vp, a.ld, lo, a.S);
    for (int64_t k0 = lo; k0 < hi; k0 += 64) {
        __syncthreads();
        stage_store<4>(regs, lds);
        if constexpr (DROP) stage_col_words<2>(bw_s, ck, k0);
        __syncthreads();
        if (k0 + 64 < hi) stage_load4(regs, kp, a.ld, vp, a.ld, k0 + 64, a.S);
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            const int64_t kb = k0 + 32 * t;
            if (kb >= hi) break;
            const char* kt = lds + t * TILE_BYTES;
            const char* vt = lds + (2 + t) * TILE_BYTES;
            f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 0), qf[0], negl, 0, 0, 0);
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vt, l31, hf, 0), dof[0], negd, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, l31, hf, 1), qf[1], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vt, l31, hf, 1), dof[1], dp, 0, 0, 0);
            if (kb + 32 > a.S) {   // wave-uniform: last tile only
                const int nv = (int)(a.S - kb);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mfma32_row(r, hf) >= nv) sc[r] = -INFINITY;
            }
            if constexpr (DROP) drop_select<false>(dp, negd, aw, bw_s + 16 * t, hf, a.drop.thr);
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = __builtin_amdgcn_exp2f(sc[r]) * dp[r];
            bf16x8 d0, d1;
            acc_to_frags(dp, d0, d1);
            dqt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(kt, lane, 0), d0, dqt, 0, 0, 0);
            dqt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(kt, lane, 1), d1, dqt, 0, 0, 0);
        }
    }
    if (qi < a.S) {
        float* dqp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + qi) * a.ld + head * D;
        const float qsc = DROP ? a.scale * a.drop.inv_keep : a.scale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(dqt[4 * g] * qsc, dqt[4 * g + 1] * qsc, dqt[4 * g + 2] * qsc, dqt[4 * g + 3] * qsc);
            *reinterpret_cast<float4*>(dqp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)qi, g, hf);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ with QB query blocks per wave (4 waves x QB x 32 queries): one LDS read of a K / V fragment serves QB
// (key tile, query block) units and the units are independent instruction streams (same idea and same
// measurements as k_attn_bwd_dkv_kb).
// ------------------------------------------------------------------------------------------------
template <int QB, int NT, int OCC, bool DROP>
__global__ __launch_bounds__(256, OCC) void k_attn_bwd_dq_kb(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * NT * TILE_BYTES];  // K tiles, then V tiles
    __shared__ __attribute__((aligned(16))) uint32_t bw_s[DROP ? 16 * NT : 4];
    uint32_t aw[QB], ck = 0;
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int head = blockIdx.x % a.H, b = blockIdx.z;   // head-fastest order: one head per XCD
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)(blockIdx.x / a.H) * (128 * QB) + wave * (32 * QB);
    const int64_t rowbase = (int64_t)b * a.S;
    const bf16_t* kp = a.qkv + rowbase * a.ld + (a.H + hkv) * D;
    const bf16_t* vp = a.qkv + rowbase * a.ld + (a.H + a.HKV + hkv) * D;
    bf16x8 qf[QB][2], dof[QB][2];
    f32x16 dqt[QB], negl[QB], negd[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t qi = q0 + 32 * qb + l31;
        const bf16_t* qp = a.qkv + (rowbase + qi) * a.ld + head * D;
        const bf16_t* dop = a.dob + (rowbase + qi) * (a.H * D) + head * D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (qi < a.S) {
                qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s + 8 * hf);
                dof[qb][s] = *reinterpret_cast<const bf16x8*>(dop + 16 * s + 8 * hf);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { qf[qb][s][j] = 0; dof[qb][s][j] = 0; }
            }
        }
        const float lse2 = (qi < a.S) ? a.lse[((int64_t)b * a.H + head) * a.S + qi] * LOG2E : INFINITY;
        const float del = (qi < a.S) ? a.delta[((int64_t)b * a.H + head) * a.S + qi] * (DROP ? a.drop.keep : 1.f) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dqt[qb][r] = 0.f; negl[qb][r] = -lse2; negd[qb][r] = -del; }
        aw[qb] = 0;
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            aw[qb] = gdrop::row_word(gdrop::row_key(seed, bh), (uint32_t)qi);
            ck = gdrop::col_key(seed, bh);
        }
    }
    uint4 regs[NT];
    stage_loadN<NT>(regs, kp, a.ld, vp, a.ld, 0, a.S);
    for (int64_t k0 = 0; k0 < a.S; k0 += 32 * NT) {
        __syncthreads();
        stage_storeN<NT>(regs, lds);
        if constexpr (DROP) stage_col_words<NT>(bw_s, ck, k0);
        __syncthreads();
        if (k0 + 32 * NT < a.S) stage_loadN<NT>(regs, kp, a.ld, vp, a.ld, k0 + 32 * NT, a.S);
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            const int64_t kb = k0 + 32 * t;
            if (kb >= a.S) break;
            const char* kt = lds + t * TILE_BYTES;
            const char* vt = lds + (NT + t) * TILE_BYTES;
            const bf16x8 ka0 = frag_rows(kt, l31, hf, 0), va0 = frag_rows(vt, l31, hf, 0);
            const bf16x8 ka1 = frag_rows(kt, l31, hf, 1), va1 = frag_rows(vt, l31, hf, 1);
            f32x16 sc[QB], dp[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                sc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qf[qb][0], negl[qb], 0, 0, 0);
                dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, dof[qb][0], negd[qb], 0, 0, 0);
                sc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qf[qb][1], sc[qb], 0, 0, 0);
                dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, dof[qb][1], dp[qb], 0, 0, 0);
            }
            if (kb + 32 > a.S) {   // wave-uniform: last tile only
                const int nv = (int)(a.S - kb);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (mfma32_row(r, hf) >= nv) sc[qb][r] = -INFINITY;
            }
            const bf16x8 kc0 = frag_cols(kt, lane, 0), kc1 = frag_cols(kt, lane, 1);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if constexpr (DROP) drop_select<false>(dp[qb], negd[qb], aw[qb], bw_s + 16 * t, hf, a.drop.thr);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[qb][r] = __builtin_amdgcn_exp2f(sc[qb][r]) * dp[qb][r];
                bf16x8 d0, d1;
                acc_to_frags(dp[qb], d0, d1);
                dqt[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc0, d0, dqt[qb], 0, 0, 0);
                dqt[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc1, d1, dqt[qb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int64_t qi = q0 + 32 * qb + l31;
        if (qi < a.S) {
            float* dqp = a.dqkv + (rowbase + qi) * a.ld + head * D;
            const float qsc = DROP ? a.scale * a.drop.inv_keep : a.scale;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(dqt[qb][4 * g] * qsc, dqt[qb][4 * g + 1] * qsc, dqt[qb][4 * g + 2] * qsc, dqt[qb][4 * g + 3] * qsc);
                *reinterpret_cast<float4*>(dqp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)qi, g, hf);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FUSED backward: dK, dV AND dQ from ONE pass over the score tiles (round 3).  The two-pass form recomputes S, the exp and
// the dropout mask in the dQ pass -- 3 MFMA units and a full softmax to deliver 1 unit of result.  Here a workgroup of
// 8 waves owns 512 keys (wave: 2 key blocks, as k_attn_bwd_dkv_kb) and streams the queries; per (query tile, key block)
// unit the wave forms S^T / dP^T, P and dS once, feeds dV / dK as before, and then
//   * parks its bf16 dS tile [key][query] in a wave-private LDS tile (four 8-byte stores) and reads it back TRANSPOSED
//     (ds_read_b64_tr_b16) as the B operand of  dQ^T[d][q] += K^T[d][key] dS^T[key][q]  -- K^T fragments come from a
//     wave-private LDS copy of the wave's own K rows; both key blocks chain into one accumulator;
//   * writes that 32 x 32 fp32 partial to its slot in LDS; between the two barriers of the NEXT stage the 512 threads sum
//     the 8 waves' slots in wave order and store the slab's dQ partial to HBM, coalesced: dqpart[b][h][slab][S][32], rounded
//     to bf16 (the fp32 sum over the slab's 512 keys is rounded once; the GEMMs that consume dq round it to bf16 again anyway).
// k_attn_dq_reduce then sums the S/512 slabs in slab order in fp32, applies scale / (1-p), rotates back (inverse RoPE) and
// writes the q columns of dqkv.  No atomics, fixed summation order: bit-reproducible.  The slab partials cost S/512 x the
// bf16 dQ bytes of HBM writes inside this kernel (hidden: it is compute bound) and one streaming pass to reduce them
// (256 MB per layer at S = 16 384, 8 heads: 0.07 ms).
// LDS: stage tiles 8 KB + row constants + K tiles 32 KB + dS tiles 32 KB + dQ slots 64 KB = 137 KB -> one workgroup per CU,
// two waves per SIMD (256 registers), grid = (S/512) * HKV workgroups: at S = 16 384, 8 heads exactly one per CU.
// ------------------------------------------------------------------------------------------------
// LDS layout of k_attn_bwd_fused<DROP, W waves, KB key blocks per wave, NT query tiles per stage>
template <int W, int KB, int NT>
struct FusedLds {
    static constexpr int QS = 32 * NT;
    static constexpr int STAGE = 0;                                  // Q tiles, then dO tiles
    static constexpr int LSE = STAGE + 2 * NT * TILE_BYTES;          // float[QS]
    static constexpr int DEL = LSE + QS * 4;                         // float[QS]
    static constexpr int AW = DEL + QS * 4;                          // uint32[2 * QS + 8]
    static constexpr int K = AW + (2 * QS + 8) * 4;                  // per wave KB K tiles
    static constexpr int DS = K + W * KB * TILE_BYTES;               // per wave KB dS tiles
    static constexpr int SLOT = DS + W * KB * TILE_BYTES;            // [NT][wave][32 q][32 d] fp32
    static constexpr int TOTAL = SLOT + NT * W * 4096;
    static constexpr int KEYS = W * KB * 32;
};

struct FusedArgs {
    BwdArgs a;
    bf16_t* dqpart;       // [B][H][nslab][S][32] bf16 (fp32 sums over the slab's keys, rounded once: see k_attn_dq_reduce)
    int nslab;
    int lab;              // measurement switch (GAOT_ATTN_BWD_LAB): selects the SB instantiation on the host side
    unsigned long long* stamps;   // ORD == 2 (diagnostic build, GAOT_ATTN_BWD_STAMPS=1): 16 cycle sums per wave, else NULL
};

// FB_WAVES x FB_KB = 16 key blocks (512 keys) per workgroup either way: 8 waves x 2 blocks run two waves per SIMD in 256
// registers each; 4 waves x 4 blocks run ONE wave per SIMD with the whole 512-register file (K^T fragments of the dQ product
// kept in registers, one read of a Q / dO fragment or row constant serves four units, four slots to reduce instead of eight).
// PK (dropout only): the row words arrive packed per PAIR of queries (the two consecutive queries of an accumulator register
// pair), so one xor makes the uniform 16-bit values of both and the two compares read its halves directly (SDWA): 2.5
// mask-generation instructions per pair instead of 4, and half the row-word reads.
// ORD (with PK): 0 = all of a tile's exp / mask work, then all of its dV / dK products; 1 = key block by key block (block 0's
// dV / dK MFMAs are in flight under block 1's exp / mask stream)
// TT (round 5): the dS tile reaches the dQ product's "query on the lane" orientation on the MATRIX pipe -- the accumulator as the
// A operand [key][query] times a permutation fragment I[query][query'] (two MFMAs, exact: products with 1.0 / 0.0), packed again --
// instead of through the wave-private LDS tile (4 ds_write_b64 + 4 ds_read_b64_tr_b16 per unit on an issue port that is 89 % busy)
template <bool DROP, int FB_WAVES, int FB_KB, int FB_NT, bool PK = false, int ORD = 0, int SB = 0, bool TT = false>
__global__ __launch_bounds__(64 * FB_WAVES, FB_WAVES == 8 ? 1 : (FB_KB >= 4 ? 1 : 2)) void k_attn_bwd_fused(FusedArgs fa) {
    using L = FusedLds<FB_WAVES, FB_KB, FB_NT>;
    constexpr int FB_QS = L::QS, FB_OFF_STAGE = L::STAGE, FB_OFF_LSE = L::LSE, FB_OFF_DEL = L::DEL, FB_OFF_AW = L::AW, FB_OFF_K = L::K;
    constexpr int FB_KEYS = L::KEYS, FB_OFF_DS = L::DS, FB_OFF_SLOT = L::SLOT;
    constexpr bool KT_REGS = FB_KB >= 4;          // the wave's K^T fragments (A operand of the dQ product) live in registers
    constexpr int NTHR = 64 * FB_WAVES;
    const BwdArgs& a = fa.a;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const float dscale = DROP ? a.drop.keep : 1.f;   // see k_attn_bwd_dkv_bf16
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.x % a.HKV, b = blockIdx.z, slab = blockIdx.x / a.HKV;   // head-fastest: one (kv) head per XCD
    const int rep = a.H / a.HKV;
    // query range of this workgroup (blockIdx.y): few heads per launch -- the heads of one rank of a sharded step -- leave
    // S/512 * heads workgroups, too few for 256 CUs; the queries are then split over P parts of `chunk` rows (a multiple of the
    // stage), every part keeps dK / dV partials of its own (dqkv + part * dqkv_part, summed in part order by k_sum_cols) and
    // writes the slab partials of ITS queries
    const int64_t q_lo = (int64_t)blockIdx.y * a.chunk, q_hi = min((int64_t)a.S, q_lo + a.chunk);
    // in-kernel stamps (ORD == 2 only: a diagnostic instantiation, never the product's): cycles between consecutive stamp
    // points, summed per wave; index = the stamp that CLOSES the interval
    unsigned long long st_acc[16], st_last = 0;
    if constexpr (ORD == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st_acc[i] = 0;
        st_last = __builtin_amdgcn_s_memtime();
    }
    auto STAMP = [&](int i) {
        if constexpr (ORD == 2) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            st_acc[i] += t - st_last;
            st_last = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // priority between the two waves of a SIMD: s_setprio flips around the S / dP MFMA cluster of every tile, both waves alike
    // (measured and removed, profiles/archive/r4_b / r4_e / r4_h: static priority for either half of the waves, priority for one half
    // inside its MFMA clusters only, one half a level higher throughout, no priority at all: within +-1.5 % at compile time;
    // RUN-TIME-conditional s_setprio splits the basic blocks around the clusters and changed the dropout kernel's schedule by
    // +30 %: the flips around the S / dP cluster are unconditional)
    // SB: bit i = __builtin_amdgcn_sched_barrier(0) at phase boundary i (1: after the S / dP cluster, 2: after the exp / mask
    // stream, 4: after the dV / dK products, 8: after the dQ products) -- pins the compiler's schedule at those points
    const int64_t key0 = (int64_t)slab * FB_KEYS + wave * (32 * FB_KB);
    const int64_t rowbase = (int64_t)b * a.S;
    char* ktile = lds + FB_OFF_K + wave * FB_KB * TILE_BYTES;
    char* dstile = lds + FB_OFF_DS + wave * FB_KB * TILE_BYTES;
    bf16x8 kf[FB_KB][2], vf[FB_KB][2];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) {
        const int64_t ki = key0 + 32 * kb + l31;
        const bf16_t* kp = a.qkv + (rowbase + ki) * a.ld + (a.H + hkv) * D;
        const bf16_t* vp = a.qkv + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (ki < a.S) {
                kf[kb][s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s + 8 * hf);
                vf[kb][s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s + 8 * hf);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kf[kb][s][j] = 0; vf[kb][s][j] = 0; }
            }
            // wave-private [key][d] copy of the K rows: its transposed fragments are the A operand of the dQ product
            *reinterpret_cast<bf16x8*>(ktile + kb * TILE_BYTES + tile_off(l31, 2 * s + hf)) = kf[kb][s];
        }
    }
    f32x16 dkt[FB_KB], dvt[FB_KB];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[kb][r] = 0.f; dvt[kb][r] = 0.f; }
    // TT: permutation fragments, B operand [k = query][n = query'], element j of k-step s stands for query 16 s + 8 (j >> 2) +
    // 4 hf + (j & 3) (the k order of an accumulator used as an operand): 1.0 where that query is the lane's own
    bf16x8 idf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[s][j] = (TT && (16 * s + 8 * (j >> 2) + 4 * hf + (j & 3)) == l31) ? (short)0x3F80 : (short)0;
    // loop-invariant K^T fragments (the A operand of dQ^T = K^T dS^T): read back once from the wave's own tile when the
    // register file has room for them (one wave per SIMD), else re-read per tile (two waves per SIMD, 256 registers)
    bf16x8 ktf[KT_REGS ? FB_KB : 1][2];
    if constexpr (KT_REGS) {
#pragma unroll
        for (int kb = 0; kb < FB_KB; ++kb) {
            ktf[kb][0] = frag_cols(ktile + kb * TILE_BYTES, lane, 0);
            ktf[kb][1] = frag_cols(ktile + kb * TILE_BYTES, lane, 1);
        }
    }

    // staging: 2 NT 128 16-byte chunks per stage (tile t of Q0.. dO0.., row r, chunk c), NST per thread
    constexpr int NSTG = NTHR;
    constexpr int NST = 2 * FB_NT * 128 / NSTG;
    // slot reduction: NT 256 (tile, query, 16-byte chunk) items per stage -> 4 consecutive d each, NRS per thread
    constexpr int NRS = FB_NT * 256 / NTHR;
    static_assert(NST >= 1 && NRS >= 1 && FB_QS <= NTHR, "stage shape vs workgroup size");

    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
        const bf16_t* dop = a.dob + rowbase * (a.H * D) + head * D;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        bf16_t* part = fa.dqpart + (((int64_t)b * a.H + head) * fa.nslab + slab) * (int64_t)a.S * D;
        // the stage's Q / dO rows through buffer resources of this (batch, head): one 32-bit lane offset per load and a scalar row
        // offset instead of 64-bit lane addresses (they were spilled: the kernel then needs scratch, and a scratch kernel costs
        // ~10 us of idle queue on either side of its launch -- profiles/archive/r4_q_gaps.txt); rows past S read as zeros in hardware
        const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (int)((int64_t)a.S * a.ld * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t do_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dop, 0, (int)((int64_t)a.S * a.H * D * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t lse_rs = __builtin_amdgcn_make_buffer_rsrc((void*)lsep, 0, a.S * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t del_rs = __builtin_amdgcn_make_buffer_rsrc((void*)delp, 0, a.S * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t part_rs = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, (int)((int64_t)a.S * D * 2), 0x00020000);
        static_assert(NSTG % 128 == 0, "a wave stages rows of one tile");
        const int st_tw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
        int st_voff[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = threadIdx.x + it * NSTG, st_t = idx >> 7, st_r = (idx & 127) >> 2, st_c = idx & 3;
            const int row = 32 * (st_t % FB_NT) + st_r;
            st_voff[it] = (st_t < FB_NT) ? row * (int)a.ld * 2 + 16 * st_c : row * (a.H * D * 2) + 16 * st_c;
        }
        auto stage_load = [&](uint4 (&rg)[NST], int64_t q0) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int st_t = st_tw + ((it * NSTG) >> 7);          // wave-uniform (a scalar branch)
                if (st_t < FB_NT) rg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, st_voff[it], (int)q0 * (int)a.ld * 2, 0));
                else rg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(do_rs, st_voff[it], (int)q0 * (a.H * D * 2), 0));
            }
        };
        auto stage_store = [&](const uint4 (&rg)[NST], char* sb) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = threadIdx.x + it * NSTG, st_t = idx >> 7, st_r = (idx & 127) >> 2, st_c = idx & 3;
                *reinterpret_cast<uint4*>(sb + FB_OFF_STAGE + st_t * TILE_BYTES + tile_off(st_r, st_c)) = rg[it];
            }
        };
        auto reduce_slots = [&](int64_t q0) {    // sum the waves' dQ^T partials of the stage that started at q0, wave order
#pragma unroll
            for (int it = 0; it < NRS; ++it) {
                const int idx = threadIdx.x + it * NTHR;
                const int rt = idx >> 8, rq = (idx >> 3) & 31, rc = idx & 7;
                const char* sp = lds + FB_OFF_SLOT + rt * FB_WAVES * 4096 + rq * 128 + ((rc ^ (rq & 7)) << 4);
                float4 acc = *reinterpret_cast<const float4*>(sp);
#pragma unroll
                for (int w = 1; w < FB_WAVES; ++w) {
                    const float4 v = *reinterpret_cast<const float4*>(sp + w * 4096);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                // rows past S fall outside the resource: the store is dropped in hardware
                __builtin_amdgcn_raw_buffer_store_b64(
                    __builtin_bit_cast(u32x2_t, make_uint2((unsigned)f2bf(acc.x) | ((unsigned)f2bf(acc.y) << 16), (unsigned)f2bf(acc.z) | ((unsigned)f2bf(acc.w) << 16))),
                    part_rs, (32 * rt + rq) * (D * 2) + 8 * rc, (int)q0 * (D * 2), 0);
            }
        };
        uint4 regs[NST];
        stage_load(regs, q_lo);
        // -lse / -delta of the stage's queries: loaded RAW (clamped address, no arithmetic on the value) one stage ahead, so
        // that the wait for them falls where they are stored -- a whole stage later -- and not right behind the load (an
        // `if (q < S) x = -lse[q] * c` puts load, s_waitcnt vmcnt(0) and multiply into one block: wave 0 then sat out two
        // dependent L2 round trips per stage, queued behind the stage's own tile loads, while seven waves waited at the barrier)
        float lt = 0.f, et = 0.f;
        auto load_consts = [&](int64_t qbase) {
            if (threadIdx.x < FB_QS) {     // rows past S: zeros in hardware (staged as -inf / 0 below)
                lt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lse_rs, threadIdx.x * 4, (int)qbase * 4, 0));
                et = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(del_rs, threadIdx.x * 4, (int)qbase * 4, 0));
            }
        };
        load_consts(q_lo);
        uint32_t rk = 0, bsel[FB_KB];
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t ck = gdrop::col_key(seed, bh);
#pragma unroll
            for (int kb = 0; kb < FB_KB; ++kb) {
                const int64_t ki = key0 + 32 * kb + l31;
                const uint32_t bw = gdrop::col_word(ck, (uint32_t)(ki >> 1));
                bsel[kb] = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
                if constexpr (PK) bsel[kb] |= bsel[kb] << 16;
            }
        }
        const uint32_t thr_v = a.drop.thr;
        // everything a stage needs besides the tiles: row constants (staged NEGATED: the initial accumulator values) and the
        // dropout row words of queries q0 ..; written by the first FB_QS (FB_QS / 2) threads into stage buffer sb
        auto stage_consts = [&](char* sb, int64_t q0) {
            float* lse_w = reinterpret_cast<float*>(sb + FB_OFF_LSE);
            float* del_w = reinterpret_cast<float*>(sb + FB_OFF_DEL);
            uint32_t* aw_w = reinterpret_cast<uint32_t*>(sb + FB_OFF_AW);
            if (threadIdx.x < FB_QS) {
                const bool in = q0 + threadIdx.x < a.S;
                lse_w[threadIdx.x] = in ? -lt * LOG2E : -INFINITY;
                del_w[threadIdx.x] = in ? -et * dscale : 0.f;
            }
            if constexpr (DROP && !PK) stage_row_words<FB_QS>(aw_w, rk, q0);
            if constexpr (DROP && PK) {
                // packed form: copy c (key parity), word ((t 2 + hf) 4 + g4) 2 + j = halfword c of the row words of queries
                // 32 t + 8 g4 + 4 hf + 2 j (low half) and + 1 (high half): a lane's 8 words of a tile are 32 contiguous bytes
                if (threadIdx.x < FB_QS / 2) {
                    const int u = threadIdx.x, st = u >> 4, sh = (u >> 3) & 1, sg = (u >> 1) & 3, sj = u & 1;
                    const uint32_t qe = (uint32_t)q0 + 32 * st + 8 * sg + 4 * sh + 2 * sj;
                    const uint32_t w0 = gdrop::row_word(rk, qe), w1 = gdrop::row_word(rk, qe + 1);
                    aw_w[u] = (w0 & 0xffffu) | (w1 << 16);
                    aw_w[FB_QS / 2 + 4 + u] = (w0 >> 16) | (w1 & 0xffff0000u);
                }
            }
        };
        char* const sbuf = lds;
        const float* lse_s = reinterpret_cast<const float*>(sbuf + FB_OFF_LSE);
        const float* del_s = reinterpret_cast<const float*>(sbuf + FB_OFF_DEL);
        const uint32_t* aw_s = reinterpret_cast<const uint32_t*>(sbuf + FB_OFF_AW);
        for (int64_t q0 = q_lo; q0 < q_hi; q0 += FB_QS) {
            {
                STAMP(0);            // end of the previous stage's tiles (incl. its slot stores)
                __syncthreads();     // A: every wave is done with the staged tiles and has written its slots of the previous stage
                STAMP(1);            // wait at barrier A
                // the tile loads issued a stage ago are waited for HERE, before this section's slab-partial store is issued (vmcnt
                // counts stores too: behind the store the wait would cover its whole round trip)
                stage_store(regs, sbuf);
                stage_consts(sbuf, q0);
                if (q0 > q_lo) reduce_slots(q0 - FB_QS);
                STAMP(2);            // staging stores + slot reduction
                __syncthreads();     // B
                STAMP(3);            // wait at barrier B
                if (q0 + FB_QS < q_hi) {
                    stage_load(regs, q0 + FB_QS);
                    load_consts(q0 + FB_QS);
                }
            }
            STAMP(4);            // issue of the next stage's global loads
#pragma unroll
            for (int t = 0; t < FB_NT; ++t) {
                if (q0 + 32 * t >= q_hi) break;
                if constexpr ((SB & 16) != 0) __builtin_amdgcn_sched_barrier(0);
                const char* qt = sbuf + FB_OFF_STAGE + t * TILE_BYTES;
                const char* dt = sbuf + FB_OFF_STAGE + (FB_NT + t) * TILE_BYTES;
                f32x16 lc, dc;
                load_row_consts(lc, lse_s + 32 * t, hf);
                load_row_consts(dc, del_s + 32 * t, hf);
                const bf16x8 qa0 = frag_rows(qt, l31, hf, 0), da0 = frag_rows(dt, l31, hf, 0);
                const bf16x8 qa1 = frag_rows(qt, l31, hf, 1), da1 = frag_rows(dt, l31, hf, 1);
                f32x16 sc[FB_KB], dp[FB_KB];
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < FB_KB; ++kb) {
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0, kf[kb][0], lc, 0, 0, 0);
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, vf[kb][0], dc, 0, 0, 0);
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1, kf[kb][1], sc[kb], 0, 0, 0);
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, vf[kb][1], dp[kb], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                if constexpr ((SB & 1) != 0) __builtin_amdgcn_sched_barrier(0);
                STAMP(5);        // row constants + row fragments read, S / dP MFMAs issued
                const bf16x8 dc0 = frag_cols(dt, lane, 0), dc1 = frag_cols(dt, lane, 1);
                const bf16x8 qc0 = frag_cols(qt, lane, 0), qc1 = frag_cols(qt, lane, 1);
                uint32_t w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (DROP && PK) {
                    const uint4* awp = reinterpret_cast<const uint4*>(aw_s + (l31 & 1) * (FB_QS / 2 + 4) + (t * 2 + hf) * 8);
                    const uint4 wa = awp[0], wb = awp[1];
                    w8[0] = wa.x; w8[1] = wa.y; w8[2] = wa.z; w8[3] = wa.w; w8[4] = wb.x; w8[5] = wb.y; w8[6] = wb.z; w8[7] = wb.w;
                }
                auto mask_pair = [&](int kb, int g4, int j) {     // two consecutive queries of key block kb: registers r0, r0 + 1
                    const int r0 = 4 * g4 + 2 * j, r1 = r0 + 1;
                    uint32_t x;
                    unsigned long long m0, m1;
                    asm("v_xor_b32 %0, %3, %4\n\t"
                        "v_cmp_ge_u32_sdwa %1, %0, %5 src0_sel:WORD_0 src1_sel:DWORD\n\t"
                        "v_cmp_ge_u32_sdwa %2, %0, %5 src0_sel:WORD_1 src1_sel:DWORD"
                        : "=&v"(x), "=&s"(m0), "=&s"(m1) : "v"(w8[2 * g4 + j]), "v"(bsel[kb]), "v"(thr_v));
                    const float p0 = __builtin_amdgcn_exp2f(sc[kb][r0]), p1 = __builtin_amdgcn_exp2f(sc[kb][r1]);
                    float pm0, pm1, t0, t1;
                    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(pm0) : "v"(p0), "s"(m0));
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t0) : "v"(dc[r0]), "v"(dp[kb][r0]), "s"(m0));
                    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(pm1) : "v"(p1), "s"(m1));
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t1) : "v"(dc[r1]), "v"(dp[kb][r1]), "s"(m1));
                    sc[kb][r0] = pm0; sc[kb][r1] = pm1;
                    dp[kb][r0] = p0 * t0; dp[kb][r1] = p1 * t1;
                };
                if constexpr (DROP && PK && ORD != 1) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                        for (int kb = 0; kb < FB_KB; ++kb)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mask_pair(kb, g4, j);
                }
                if constexpr ((SB & 2) != 0) __builtin_amdgcn_sched_barrier(0);
                STAMP(6);        // wait for the MFMA results + exp / mask stream (PK)
                if constexpr (DROP && !PK) {   // row words read 4 at a time: the two key blocks share them (measured: key-block-outer order,
                    // which would put block 0's MFMAs under block 1's mask work, is 7 % slower)
                    const uint32_t* awp = aw_s + (l31 & 1) * (FB_QS + 4) + 32 * t + 4 * hf;
                    // the row words of group g4 + 1 are requested before group g4 is worked on (left alone the compiler loads
                    // each group right before its first use and the wave sits out the LDS round trip four times per tile)
                    uint4 wv = *reinterpret_cast<const uint4*>(awp);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint4 nxt = wv;
                        if (g4 < 3) {
                            nxt = *reinterpret_cast<const uint4*>(awp + 8 * (g4 + 1));
                            asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w));   // keep the load here
                        }
                        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int kb = 0; kb < FB_KB; ++kb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r = 4 * g4 + i;
                                const float p = __builtin_amdgcn_exp2f(sc[kb][r]);
                                const bool keep = (ww[i] ^ bsel[kb]) >= a.drop.thr;
                                float pm = keep ? p : 0.f;
                                asm volatile("" : "+v"(pm));   // select on fp32, then ONE cvt_pk per pair (see k_attn_bwd_dkv_kb)
                                sc[kb][r] = pm;
                                dp[kb][r] = p * (keep ? dp[kb][r] : dc[r]);
                            }
                        wv = nxt;
                    }
                }
                bf16x8 tsf[FB_KB][2];     // TT: the unit's dS^T fragments (query on the lane)
#pragma unroll
                for (int kb = 0; kb < FB_KB; ++kb) {
                    if constexpr (DROP && PK && ORD == 1) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mask_pair(kb, g4, j);
                    }
                    if constexpr (!DROP) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float p = __builtin_amdgcn_exp2f(sc[kb][r]);   // rows beyond S carry lse = +inf -> p = 0
                            sc[kb][r] = p;
                            dp[kb][r] = p * dp[kb][r];
                        }
                    }
                    bf16x8 p0, p1, d0, d1;
                    acc_to_frags(sc[kb], p0, p1);
                    acc_to_frags(dp[kb], d0, d1);
                    dvt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dc0, p0, dvt[kb], 0, 0, 0);
                    dkt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qc0, d0, dkt[kb], 0, 0, 0);
                    dvt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dc1, p1, dvt[kb], 0, 0, 0);
                    dkt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qc1, d1, dkt[kb], 0, 0, 0);
                    if constexpr (TT) {
                        f32x16 tr;
#pragma unroll
                        for (int r = 0; r < 16; ++r) tr[r] = 0.f;
                        tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, idf[0], tr, 0, 0, 0);
                        tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, idf[1], tr, 0, 0, 0);
                        acc_to_frags(tr, tsf[kb][0], tsf[kb][1]);
                    } else {
                    // dS tile [key = l31][query]: the lane's 4 runs of 4 consecutive queries (8 g + 4 hf + 0..3) = 8-byte stores
                    const uint4 lo = __builtin_bit_cast(uint4, d0), hi = __builtin_bit_cast(uint4, d1);
                    char* dst = dstile + kb * TILE_BYTES;
                    *reinterpret_cast<uint2*>(dst + tile_off(l31, 0) + 8 * hf) = make_uint2(lo.x, lo.y);
                    *reinterpret_cast<uint2*>(dst + tile_off(l31, 1) + 8 * hf) = make_uint2(lo.z, lo.w);
                    *reinterpret_cast<uint2*>(dst + tile_off(l31, 2) + 8 * hf) = make_uint2(hi.x, hi.y);
                    *reinterpret_cast<uint2*>(dst + tile_off(l31, 3) + 8 * hf) = make_uint2(hi.z, hi.w);
                    }
                }
                if constexpr ((SB & 4) != 0) __builtin_amdgcn_sched_barrier(0);
                STAMP(7);        // conversions, dV / dK MFMAs issued, dS tile stored
                // dQ^T[d][q] of this wave's 64 keys: K^T (A, transposed read of the K tile) x dS^T (B, transposed read of
                // the tile just written: same LDS object, so the compiler keeps the order, and LDS operations of one wave
                // complete in order)
                f32x16 dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < FB_KB; ++kb) {
                    if constexpr (TT) {
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KT_REGS ? ktf[kb][0] : frag_cols(ktile + kb * TILE_BYTES, lane, 0), tsf[kb][0], dq, 0, 0, 0);
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KT_REGS ? ktf[kb][1] : frag_cols(ktile + kb * TILE_BYTES, lane, 1), tsf[kb][1], dq, 0, 0, 0);
                    } else if constexpr (KT_REGS) {
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[kb][0], frag_cols(dstile + kb * TILE_BYTES, lane, 0), dq, 0, 0, 0);
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[kb][1], frag_cols(dstile + kb * TILE_BYTES, lane, 1), dq, 0, 0, 0);
                    } else {
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(ktile + kb * TILE_BYTES, lane, 0),
                                                                     frag_cols(dstile + kb * TILE_BYTES, lane, 0), dq, 0, 0, 0);
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(ktile + kb * TILE_BYTES, lane, 1),
                                                                     frag_cols(dstile + kb * TILE_BYTES, lane, 1), dq, 0, 0, 0);
                    }
                }
                // slot [q = l31][32 d] fp32, 16-byte chunk index (2 g + hf) XOR (q & 7): conflict-free stores and reduction reads
                if constexpr ((SB & 8) != 0) __builtin_amdgcn_sched_barrier(0);
                STAMP(8);        // dS round trip through LDS + dQ MFMAs issued
                char* slot = lds + FB_OFF_SLOT + (t * FB_WAVES + wave) * 4096 + l31 * 128;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(slot + (((2 * g + hf) ^ (l31 & 7)) << 4)) =
                        make_float4(dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]);
                STAMP(9);        // wait for the dQ MFMAs + slot stores
            }
        }
        __syncthreads();
        reduce_slots(q_lo + ((q_hi - 1 - q_lo) / FB_QS) * (int64_t)FB_QS);
    }
    if constexpr (ORD == 2) {
        if (fa.stamps && lane == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) fa.stamps[((int64_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * FB_WAVES + wave) * 16 + i] = st_acc[i];
        }
    }
    const float vsc = DROP ? a.drop.inv_keep : 1.f;
    const float ksc = vsc / LOG2E;
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) {
        const int64_t ki = key0 + 32 * kb + l31;
        if (ki < a.S) {
            float* dkp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + hkv) * D;
            float* dvp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(dkt[kb][4 * g] * ksc, dkt[kb][4 * g + 1] * ksc, dkt[kb][4 * g + 2] * ksc, dkt[kb][4 * g + 3] * ksc);
                *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)ki, g, hf);
                float4 u = make_float4(dvt[kb][4 * g] * vsc, dvt[kb][4 * g + 1] * vsc, dvt[kb][4 * g + 2] * vsc, dvt[kb][4 * g + 3] * vsc);
                *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_attn_bwd_asm: the same fused pass with ONE wave per SIMD and a hand-scheduled tile loop.  Workgroup = 4 waves x
// 4 key blocks = 512 keys (same slabs, same dQ slab partials and reduction, same bit-reproducibility), stages of 128 queries.
// The stage code (global loads, LDS staging, slot reduction, barriers, epilogue) is the C++ below; the tile loop of a stage -- 16
// (query tile, key block) units = 160 MFMAs, 1 724 vector instructions -- is ONE generated asm statement
// (gen_attn_bwd_asm.py -> attn_bwd_asm.inc) that owns v48-v255, s64-s95 and the AGPRs: a0-a127 dK^T / dV^T accumulators,
// a128-a191 K / V row fragments, a192-a223 K^T fragments.  The AGPRs live ACROSS the asm statements: nothing else in this kernel
// touches them (no MFMA intrinsic, no spill -- tests/test_host_cpu.py checks the code object for v_accvgpr outside the asm's own).
// ------------------------------------------------------------------------------------------------
#include "attn_bwd_asm.inc"
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// four consecutive AGPRs a[BASE .. BASE+3] <- the 16 bytes of v (the register numbers are part of the instruction text)
template <int BASE>
__device__ __forceinline__ void agpr_write4(uint4 v) {
    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%4+1], %1\n\t"
                 "v_accvgpr_write_b32 a[%4+2], %2\n\tv_accvgpr_write_b32 a[%4+3], %3"
                 :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "n"(BASE) : "memory");
}
template <int REG>
__device__ __forceinline__ float agpr_read() {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(REG) : "memory");
    return r;
}
template <bool MT>     // MT: the dS tile is transposed on the matrix pipe (no wave-private dS tiles in LDS)
struct AsmLdsT {
    static constexpr int NT = GAOT_ATTN_BWD_ASM_NT, W = 4, KB = 4, QS = 32 * NT;
    static constexpr int STAGE = 0;                                  // Q tiles, then dO tiles
    static constexpr int LSE = STAGE + 2 * NT * TILE_BYTES;          // float[QS]
    static constexpr int DEL = LSE + QS * 4;                         // float[QS]
    static constexpr int AW = DEL + QS * 4;                          // uint32[QS + 8] packed row words (two parity copies)
    static constexpr int DS = (AW + (QS + 8) * 4 + 2047) / 2048 * 2048;   // per wave KB dS tiles (2 KB aligned: xor addressing)
    static constexpr int SLOT = DS + (MT ? 0 : W * KB * TILE_BYTES);   // [NT][wave][32 q][32 d] fp32; the waves' K tiles alias it at start
    static constexpr int TOTAL = SLOT + NT * W * 4096;
    static constexpr int KEYS = W * KB * 32;
};
template <bool DROP>
using AsmLds = AsmLdsT<DROP ? (GAOT_ATTN_BWD_ASM_MFMA_T_DROP != 0) : (GAOT_ATTN_BWD_ASM_MFMA_T_NODROP != 0)>;
static_assert(AsmLds<false>::TOTAL <= 160 * 1024 && AsmLds<true>::SLOT % 128 == 0 && AsmLds<false>::SLOT % 128 == 0 &&
              AsmLds<true>::NT * 4 * 4096 >= 4 * 4 * TILE_BYTES, "LDS layout");

// LAB (measurement only, results invalid): 1 = the stage code without the tile loop, 2 = the tile loop without slot reduction /
// row-word hashing (barriers and tile staging kept), 3 = no barriers either
template <bool DROP, int LAB = 0>
__global__ __launch_bounds__(256, 1) void k_attn_bwd_asm(FusedArgs fa) {
    using L = AsmLds<DROP>;
    constexpr bool MT = DROP ? (GAOT_ATTN_BWD_ASM_MFMA_T_DROP != 0) : (GAOT_ATTN_BWD_ASM_MFMA_T_NODROP != 0);
    constexpr int NT = L::NT, KB = L::KB, QS = L::QS, NTHR = 256, WV = L::W;
    const BwdArgs& a = fa.a;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const float dscale = DROP ? a.drop.keep : 1.f;
    unsigned long long seed = 0;
    if constexpr (DROP) seed = *a.drop.seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int hkv = blockIdx.x % a.HKV, b = blockIdx.z, slab = blockIdx.x / a.HKV;
    const int rep = a.H / a.HKV;
    const int64_t key0 = (int64_t)slab * L::KEYS + wave * (32 * KB);
    const int64_t rowbase = (int64_t)b * a.S;
    // query range of this workgroup (blockIdx.y): few heads per launch -- the heads of one rank of a sharded step -- split the queries
    // over parts exactly as k_attn_bwd_fused does (chunk % 128 == 0); a part's dK / dV go to its own copy (summed by k_sum_cols),
    // its dQ slab partials cover its own queries
    const int64_t q_lo = (int64_t)blockIdx.y * a.chunk, q_hi = min((int64_t)a.S, q_lo + a.chunk);
    // ---- the wave's K / V rows -> AGPR fragments; K rows once through a wave-private tile for the transposed (K^T) fragments ----
    asm volatile(GAOT_ATTN_BWD_ASM_ZERO_ACC ::: GAOT_ATTN_BWD_ASM_ACC_CLOBBERS);
    {
        char* ktile = lds + L::SLOT + wave * KB * TILE_BYTES;
        static_for<0, KB>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            const int64_t ki = key0 + 32 * kb + l31;
            const bf16_t* kp = a.qkv + (rowbase + ki) * a.ld + (a.H + hkv) * D;
            const bf16_t* vp = a.qkv + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
            static_for<0, 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                uint4 kq = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
                if (ki < a.S) {
                    kq = *reinterpret_cast<const uint4*>(kp + 16 * s + 8 * hf);
                    vq = *reinterpret_cast<const uint4*>(vp + 16 * s + 8 * hf);
                }
                *reinterpret_cast<uint4*>(ktile + kb * TILE_BYTES + tile_off(l31, 2 * s + hf)) = kq;
                agpr_write4<128 + 8 * kb + 4 * s>(kq);
                agpr_write4<160 + 8 * kb + 4 * s>(vq);
            });
        });
        static_for<0, KB>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            static_for<0, 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                agpr_write4<192 + 8 * kb + 4 * s>(__builtin_bit_cast(uint4, frag_cols(ktile + kb * TILE_BYTES, lane, s)));
            });
        });
    }
    if constexpr (MT) {
    // permutation ("identity") fragments of the dS transpose (gen_attn_bwd_asm.py: TR = "mfma"): B operand [k = query][n = query'],
    // lane (n = l31, hf), k-step s, element j stands for query 16 s + 8 (j >> 2) + 4 hf + (j & 3) -- the k order of an accumulator
    // used as an operand -- and is 1.0 where that query is the lane's own
    static_for<0, 2>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        unsigned wds[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int q0 = 16 * s + 8 * ((2 * jj) >> 2) + 4 * hf + ((2 * jj) & 3), q1 = 16 * s + 8 * ((2 * jj + 1) >> 2) + 4 * hf + ((2 * jj + 1) & 3);
            wds[jj] = (q0 == l31 ? 0x3F80u : 0u) | (q1 == l31 ? 0x3F800000u : 0u);
        }
        agpr_write4<224 + 4 * s>(make_uint4(wds[0], wds[1], wds[2], wds[3]));
    });
    }
    // ---- per-lane LDS addresses of the tile loop (32-bit LDS byte addresses) -------------------------------------------------------
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int sw = (l31 >> 2) & 3;
    const unsigned a_const = lbase + L::LSE + 16 * hf;
    const unsigned a_r0 = lbase + L::STAGE + tile_off(l31, hf), a_r1 = lbase + L::STAGE + tile_off(l31, 2 + hf);
    unsigned a_c0, a_c1;
    {
        const int i = lane & 15, grp = (lane >> 4) & 1, col = 16 * grp + 4 * (i & 3), r0 = 4 * hf + (i >> 2), r1 = r0 + 8;
        a_c0 = tile_off(r0, col >> 3) + ((col & 7) << 1);
        a_c1 = tile_off(r1, col >> 3) + ((col & 7) << 1);
    }
    // (LDS transpose only; unused -- and removed by the compiler -- when the tile is transposed on the matrix pipe)
    const unsigned a_dc0 = lbase + L::DS + wave * KB * TILE_BYTES + a_c0, a_dc1 = lbase + L::DS + wave * KB * TILE_BYTES + a_c1;
    const unsigned a_ds = lbase + L::DS + wave * KB * TILE_BYTES + l31 * 64 + (sw << 4) + 8 * hf;
    a_c0 += lbase + L::STAGE;
    a_c1 += lbase + L::STAGE;
    const unsigned a_w = lbase + L::AW + ((l31 & 1) * (QS / 2 + 4) + hf * 8) * 4;
    const unsigned a_slot = lbase + L::SLOT + wave * 4096 + l31 * 128 + ((hf ^ (l31 & 7)) << 4);

    constexpr int NST = 2 * NT * 128 / NTHR;      // staged 16-byte chunks per thread
    constexpr int NRS = NT * 256 / NTHR;          // slot-reduction items per thread
    for (int hr = 0; hr < rep; ++hr) {
        const int head = hkv * rep + hr;
        const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
        const bf16_t* dop = a.dob + rowbase * (a.H * D) + head * D;
        const float* lsep = a.lse + ((int64_t)b * a.H + head) * a.S;
        const float* delp = a.delta + ((int64_t)b * a.H + head) * a.S;
        bf16_t* part = fa.dqpart + (((int64_t)b * a.H + head) * fa.nslab + slab) * (int64_t)a.S * D;
        const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (int)((int64_t)a.S * a.ld * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t do_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dop, 0, (int)((int64_t)a.S * a.H * D * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t lse_rs = __builtin_amdgcn_make_buffer_rsrc((void*)lsep, 0, a.S * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t del_rs = __builtin_amdgcn_make_buffer_rsrc((void*)delp, 0, a.S * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t part_rs = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, (int)((int64_t)a.S * D * 2), 0x00020000);
        const int st_tw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
        int st_voff[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = threadIdx.x + it * NTHR, st_t = idx >> 7, st_r = (idx & 127) >> 2, st_c = idx & 3;
            const int row = 32 * (st_t % NT) + st_r;
            st_voff[it] = (st_t < NT) ? row * (int)a.ld * 2 + 16 * st_c : row * (a.H * D * 2) + 16 * st_c;
        }
        auto stage_load = [&](uint4 (&rg)[NST], int64_t q0) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int st_t = st_tw + ((it * NTHR) >> 7);          // wave-uniform
                if (st_t < NT) rg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, st_voff[it], (int)q0 * (int)a.ld * 2, 0));
                else rg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(do_rs, st_voff[it], (int)q0 * (a.H * D * 2), 0));
            }
        };
        auto stage_store = [&](const uint4 (&rg)[NST]) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = threadIdx.x + it * NTHR, st_t = idx >> 7, st_r = (idx & 127) >> 2, st_c = idx & 3;
                *reinterpret_cast<uint4*>(lds + L::STAGE + st_t * TILE_BYTES + tile_off(st_r, st_c)) = rg[it];
            }
        };
        auto reduce_slots = [&](int64_t q0) {    // sum the 4 waves' dQ^T partials of the stage that started at q0, wave order
#pragma unroll
            for (int it = 0; it < NRS; ++it) {
                const int idx = threadIdx.x + it * NTHR;
                const int rt = idx >> 8, rq = (idx >> 3) & 31, rc = idx & 7;
                const char* sp = lds + L::SLOT + rt * WV * 4096 + rq * 128 + ((rc ^ (rq & 7)) << 4);
                float4 acc = *reinterpret_cast<const float4*>(sp);
#pragma unroll
                for (int w = 1; w < WV; ++w) {
                    const float4 v = *reinterpret_cast<const float4*>(sp + w * 4096);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                __builtin_amdgcn_raw_buffer_store_b64(
                    __builtin_bit_cast(u32x2_t, make_uint2((unsigned)f2bf(acc.x) | ((unsigned)f2bf(acc.y) << 16), (unsigned)f2bf(acc.z) | ((unsigned)f2bf(acc.w) << 16))),
                    part_rs, (32 * rt + rq) * (D * 2) + 8 * rc, (int)q0 * (D * 2), 0);
            }
        };
        uint4 regs[NST];
        stage_load(regs, q_lo);
        float lt = 0.f, et = 0.f;
        auto load_consts = [&](int64_t qbase) {
            if (threadIdx.x < QS) {
                lt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lse_rs, threadIdx.x * 4, (int)qbase * 4, 0));
                et = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(del_rs, threadIdx.x * 4, (int)qbase * 4, 0));
            }
        };
        load_consts(q_lo);
        uint32_t rk = 0, bsel[KB] = {0, 0, 0, 0};
        if constexpr (DROP) {
            const int bh = a.drop.bh(b, head);
            rk = gdrop::row_key(seed, bh);
            const uint32_t ck = gdrop::col_key(seed, bh);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int64_t ki = key0 + 32 * kb + l31;
                const uint32_t bw = gdrop::col_word(ck, (uint32_t)(ki >> 1));
                bsel[kb] = (ki & 1) ? (bw >> 16) : (bw & 0xffffu);
                bsel[kb] |= bsel[kb] << 16;
            }
        }
        const uint32_t thr_v = a.drop.thr;
        float* lse_w = reinterpret_cast<float*>(lds + L::LSE);
        float* del_w = reinterpret_cast<float*>(lds + L::DEL);
        uint32_t* aw_w = reinterpret_cast<uint32_t*>(lds + L::AW);
        for (int64_t q0 = q_lo; q0 < q_hi; q0 += QS) {
            if constexpr (LAB != 3) __syncthreads();     // A: every wave is done with the staged tiles and has written its slots of the previous stage
            stage_store(regs);
            if (threadIdx.x < QS) {
                const bool in = q0 + threadIdx.x < a.S;
                lse_w[threadIdx.x] = in ? -lt * LOG2E : -INFINITY;
                del_w[threadIdx.x] = in ? -et * dscale : 0.f;
            }
            if constexpr (DROP && LAB < 2) {
                if (threadIdx.x < QS / 2) {   // packed row words: see k_attn_bwd_fused (PK)
                    const int u = threadIdx.x, st = u >> 4, sh = (u >> 3) & 1, sg = (u >> 1) & 3, sj = u & 1;
                    const uint32_t qe = (uint32_t)q0 + 32 * st + 8 * sg + 4 * sh + 2 * sj;
                    const uint32_t w0 = gdrop::row_word(rk, qe), w1 = gdrop::row_word(rk, qe + 1);
                    aw_w[u] = (w0 & 0xffffu) | (w1 << 16);
                    aw_w[QS / 2 + 4 + u] = (w0 >> 16) | (w1 & 0xffff0000u);
                }
            }
            if (q0 > q_lo && LAB < 2) reduce_slots(q0 - QS);
            if constexpr (LAB != 3) __syncthreads();     // B
            if (q0 + QS < q_hi) {
                stage_load(regs, q0 + QS);
                load_consts(q0 + QS);
            }
            if constexpr (LAB == 1) continue;
            if constexpr (DROP)
                asm volatile(GAOT_ATTN_BWD_STAGE_ASM_DROP
                             :: [a_const] "v"(a_const), [a_r0] "v"(a_r0), [a_r1] "v"(a_r1), [a_c0] "v"(a_c0), [a_c1] "v"(a_c1), [a_w] "v"(a_w),
#if !GAOT_ATTN_BWD_ASM_MFMA_T_DROP
                                [a_ds] "v"(a_ds), [a_dc0] "v"(a_dc0), [a_dc1] "v"(a_dc1),
#endif
                                [a_slot] "v"(a_slot), [bsel0] "v"(bsel[0]),
                                [bsel1] "v"(bsel[1]), [bsel2] "v"(bsel[2]), [bsel3] "v"(bsel[3]), [thr] "s"(thr_v)
                             : GAOT_ATTN_BWD_STAGE_ASM_CLOBBERS);
            else
                asm volatile(GAOT_ATTN_BWD_STAGE_ASM_NODROP
                             :: [a_const] "v"(a_const), [a_r0] "v"(a_r0), [a_r1] "v"(a_r1), [a_c0] "v"(a_c0), [a_c1] "v"(a_c1),
#if !GAOT_ATTN_BWD_ASM_MFMA_T_NODROP
                                [a_ds] "v"(a_ds), [a_dc0] "v"(a_dc0), [a_dc1] "v"(a_dc1),
#endif
                                [a_slot] "v"(a_slot)
                             : GAOT_ATTN_BWD_STAGE_ASM_CLOBBERS);
        }
        __syncthreads();
        reduce_slots(q_lo + ((q_hi - 1 - q_lo) / QS) * (int64_t)QS);
    }
    const float vsc = DROP ? a.drop.inv_keep : 1.f;
    const float ksc = vsc / LOG2E;
    static_for<0, KB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        const int64_t ki = key0 + 32 * kb + l31;
        float dk[16], dv[16];
        static_for<0, 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            dk[r] = agpr_read<16 * kb + r>();
            dv[r] = agpr_read<64 + 16 * kb + r>();
        });
        if (ki < a.S) {
            float* dkp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + hkv) * D;
            float* dvp = a.dqkv + blockIdx.y * a.dqkv_part + (rowbase + ki) * a.ld + (a.H + a.HKV + hkv) * D;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(dk[4 * g] * ksc, dk[4 * g + 1] * ksc, dk[4 * g + 2] * ksc, dk[4 * g + 3] * ksc);
                *reinterpret_cast<float4*>(dkp + 8 * g + 4 * hf) = unrope4(t, a.freqs, (int)ki, g, hf);
                float4 u = make_float4(dv[4 * g] * vsc, dv[4 * g + 1] * vsc, dv[4 * g + 2] * vsc, dv[4 * g + 3] * vsc);
                *reinterpret_cast<float4*>(dvp + 8 * g + 4 * hf) = u;
            }
        }
    });
}

// ------------------------------------------------------------------------------------------------
// k_attn_fwd_asm: the bound-based forward tile (k_attn_fwd_bf16<4, 4, DROP, true, 8>) with ONE wave per SIMD and a hand-scheduled
// tile loop (gen_attn_fwd_asm.py -> attn_fwd_asm.inc).  Workgroup = 4 waves x 4 query tiles = 512 queries of one head; the keys
// stream through LDS in stages of 128 (the compiled kernel's staging, two barriers per stage), and the tile loop of a stage --
// 16 (key tile, query tile) units = 64 MFMAs, 1 024 vector instructions with dropout -- is ONE asm statement.  It owns v48-v152 and
// the AGPRs a32-a159 (the four O^T accumulators and the four row-sum accumulators: two more MFMAs per unit against a fragment of
// ones), reads a0-a31 (the Q fragments); the AGPRs live ACROSS the statements.  Same arithmetic as the compiled tile (same packs, same mask words, row sums of the packed p); the
// K / V^T fragments and the mask's column words of a key tile are read from LDS once for the four query tiles.
// Few-head launches split the key range over blockIdx.y exactly as the compiled kernel does (parts combined by k_attn_combine).
// Launches whose sequence length is not a multiple of 512 (or whose key ranges are not multiples of 256) keep the compiled kernel.
// ------------------------------------------------------------------------------------------------
#include "attn_fwd_asm.inc"
struct FwdAsmLds {
    static constexpr int QT = GAOT_ATTN_FWD_ASM_QT, KT = GAOT_ATTN_FWD_ASM_KT;
    static constexpr int STAGE = 0;                              // K tiles, then V tiles
    static constexpr int BW = STAGE + 2 * KT * TILE_BYTES;       // uint32[16 * KT] column words of the stage
    static constexpr int TOTAL = BW + 16 * KT * 4;
    static constexpr int QUERIES = 4 * QT * 32, KEYS = KT * 32;
};
template <bool DROP, int LAB = 0>     // LAB 1 (measurement only, results invalid): the stage code without the tile loop
__global__ __launch_bounds__(256, 1) void k_attn_fwd_asm(FwdArgs a) {
    using L = FwdAsmLds;
    constexpr int QT = L::QT, KT = L::KT;
    __shared__ __attribute__((aligned(1024))) char lds[L::TOTAL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int head = blockIdx.x % a.H, b = blockIdx.z, qblk = blockIdx.x / a.H;
    const int hkv = head / (a.H / a.HKV);
    const int64_t q0 = (int64_t)qblk * L::QUERIES + wave * (QT * 32);
    const int64_t rowbase = (int64_t)b * a.S;
    const bf16_t* qp = a.qkv + rowbase * a.ld + head * D;
    const bf16_t* kp = a.qkv + rowbase * a.ld + (a.H + hkv) * D;
    const bf16_t* vp = a.qkv + rowbase * a.ld + (a.H + a.HKV + hkv) * D;

    // ---- the wave's Q rows -> AGPR fragments (B operand of S^T = K Q^T); |q|^2 of every row for the static bound ----------------
    float q2max = 0.f;
    static_for<0, QT>([&](auto qc) {
        constexpr int qt = decltype(qc)::value;
        const int64_t qi = q0 + 32 * qt + l31;              // S % 512 == 0: always a row of the sequence
        float q2 = 0.f;
        static_for<0, 2>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const uint4 f = *reinterpret_cast<const uint4*>(qp + qi * a.ld + 16 * s + 8 * hf);
            agpr_write4<8 * qt + 4 * s>(f);
            const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
                q2 = fmaf(lo, lo, fmaf(hi, hi, q2));
            }
        });
        q2 += xhalf(q2);
        q2max = fmaxf(q2max, q2);
    });
    {
        constexpr float BOUND2 = 54.f * 54.f;               // see k_attn_fwd_bf16 (FAST)
        float k2 = a.kmax2[((int64_t)b * KN_BLOCKS + lane) * a.HKV + hkv];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k2 = fmaxf(k2, __shfl_xor(k2, o, 64));
        const int over = __syncthreads_or(!(q2max * k2 <= BOUND2));
        // the adaptive kernel runs on the compiled kernel's grid (128 queries per workgroup): this workgroup stands for four of them
        // (blockIdx.y = key-range part of a few-head launch: the compiled grid's y, same flag layout)
        if (threadIdx.x < 4)
            a.redo[(qblk * 4 + threadIdx.x) * a.H + head + (int64_t)(a.S / 128) * a.H * (blockIdx.y + gridDim.y * b)] = over ? 1 : 0;
        if (over) return;
    }
    uint32_t aw[QT] = {0, 0, 0, 0}, ck = 0;
    if constexpr (DROP) {
        const unsigned long long seed = *a.drop.seed;
        const int bh = a.drop.bh(b, head);
        const uint32_t rk = gdrop::row_key(seed, bh);
        ck = gdrop::col_key(seed, bh);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) aw[qt] = gdrop::row_word(rk, (uint32_t)(q0 + 32 * qt + l31)) ^ 0x80008000u;
    }
    const uint32_t ts = (uint32_t)((int)a.drop.thr - 1 - 32768) & 0xffffu;
    const uint32_t tpk = ts | (ts << 16);
    asm volatile(GAOT_ATTN_FWD_ASM_ZERO_ACC ::: GAOT_ATTN_FWD_ASM_ACC_CLOBBERS);
    float l0[QT] = {0.f, 0.f, 0.f, 0.f}, l1[QT] = {0.f, 0.f, 0.f, 0.f};

    // ---- per-lane LDS addresses of the tile loop (32-bit LDS byte addresses; tile offsets are immediates of the loop) ----------------
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned a_k0 = lbase + L::STAGE + tile_off(l31, hf), a_k1 = lbase + L::STAGE + tile_off(l31, 2 + hf);
    unsigned a_v0, a_v1;
    {
        const int i = lane & 15, grp = (lane >> 4) & 1, col = 16 * grp + 4 * (i & 3), r0 = 4 * hf + (i >> 2), r1 = r0 + 8;
        a_v0 = lbase + L::STAGE + tile_off(r0, col >> 3) + ((col & 7) << 1);
        a_v1 = lbase + L::STAGE + tile_off(r1, col >> 3) + ((col & 7) << 1);
    }
    const unsigned a_w = lbase + L::BW + hf * 32;
    uint32_t* bw_s = reinterpret_cast<uint32_t*>(lds + L::BW);

    // keys [lo, hi) of this workgroup: the whole sequence, or part blockIdx.y of a few-head launch (chunk % 256 == 0; the parts'
    // normalised O / lse are combined by k_attn_combine as for the compiled kernel)
    const int64_t lo = (int64_t)blockIdx.y * a.chunk, hi = min((int64_t)a.S, lo + a.chunk);
    uint4 regs[KT];
    stage_loadN<KT>(regs, kp, a.ld, vp, a.ld, lo, a.S);
    for (int64_t k0 = lo; k0 < hi; k0 += L::KEYS) {
        __syncthreads();
        stage_storeN<KT>(regs, lds + L::STAGE);
        if constexpr (DROP) stage_col_words<KT>(bw_s, ck, k0);
        __syncthreads();
        if (k0 + L::KEYS < hi) stage_loadN<KT>(regs, kp, a.ld, vp, a.ld, k0 + L::KEYS, a.S);
        if constexpr (LAB == 1) continue;
#if GAOT_ATTN_FWD_ASM_LSUM_MFMA
        // (the row sums accumulate in a96-a159 through two MFMAs per unit against a fragment of ones)
        if constexpr (DROP)
            asm volatile(GAOT_ATTN_FWD_STAGE_ASM_DROP
                         :: [a_k0] "v"(a_k0), [a_k1] "v"(a_k1), [a_v0] "v"(a_v0), [a_v1] "v"(a_v1), [a_w] "v"(a_w),
                            [aw0] "v"(aw[0]), [aw1] "v"(aw[1]), [aw2] "v"(aw[2]), [aw3] "v"(aw[3]), [tpk] "s"(tpk)
                         : GAOT_ATTN_FWD_STAGE_ASM_CLOBBERS);
        else
            asm volatile(GAOT_ATTN_FWD_STAGE_ASM_NODROP
                         :: [a_k0] "v"(a_k0), [a_k1] "v"(a_k1), [a_v0] "v"(a_v0), [a_v1] "v"(a_v1)
                         : GAOT_ATTN_FWD_STAGE_ASM_CLOBBERS);
#else
        if constexpr (DROP)
            asm volatile(GAOT_ATTN_FWD_STAGE_ASM_DROP
                         : [l00] "+v"(l0[0]), [l01] "+v"(l0[1]), [l02] "+v"(l0[2]), [l03] "+v"(l0[3]),
                           [l10] "+v"(l1[0]), [l11] "+v"(l1[1]), [l12] "+v"(l1[2]), [l13] "+v"(l1[3])
                         : [a_k0] "v"(a_k0), [a_k1] "v"(a_k1), [a_v0] "v"(a_v0), [a_v1] "v"(a_v1), [a_w] "v"(a_w),
                           [aw0] "v"(aw[0]), [aw1] "v"(aw[1]), [aw2] "v"(aw[2]), [aw3] "v"(aw[3]), [tpk] "s"(tpk)
                         : GAOT_ATTN_FWD_STAGE_ASM_CLOBBERS);
        else
            asm volatile(GAOT_ATTN_FWD_STAGE_ASM_NODROP
                         : [l00] "+v"(l0[0]), [l01] "+v"(l0[1]), [l02] "+v"(l0[2]), [l03] "+v"(l0[3]),
                           [l10] "+v"(l1[0]), [l11] "+v"(l1[1]), [l12] "+v"(l1[2]), [l13] "+v"(l1[3])
                         : [a_k0] "v"(a_k0), [a_k1] "v"(a_k1), [a_v0] "v"(a_v0), [a_v1] "v"(a_v1)
                         : GAOT_ATTN_FWD_STAGE_ASM_CLOBBERS);
#endif
    }
    // ---- epilogue: o = acc / l (and 1 / (1 - p)), lse = log l (the bound-based tile carries no reference value) --------------------
    static_for<0, QT>([&](auto qc) {
        constexpr int qt = decltype(qc)::value;
#if GAOT_ATTN_FWD_ASM_LSUM_MFMA
        const float l = agpr_read<96 + 16 * qt>();          // every row of the ones-product is the complete row sum of the lane's query
        (void)l0; (void)l1;
#else
        float l = l0[qt] + l1[qt];
        l += xhalf(l);
#endif
        const float inv = DROP ? a.drop.inv_keep / l : 1.f / l;
        float o[16];
        static_for<0, 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            o[r] = agpr_read<32 + 16 * qt + r>() * inv;
        });
        const int64_t qi = q0 + 32 * qt + l31;
        float* op = a.o + blockIdx.y * a.o_part + (rowbase + qi) * (a.H * D) + head * D;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hf) = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
        if (hf == 0) a.lse[blockIdx.y * a.lse_part + ((int64_t)b * a.H + head) * a.S + qi] = logf(l);
    });
}

// dq[b, q, head, :] = scale / (1-p) * sum over the key slabs (slab order) of the fused kernel's partials, rotated back
__global__ void k_attn_dq_reduce(const bf16_t* __restrict__ part, int nslab, int B, int S, int H, int ld, float qsc,
                                 const float* __restrict__ freqs, float* __restrict__ dqkv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b, head, q, 4-column group)
    const int64_t n = (int64_t)B * H * S * 8;
    if (i >= n) return;
    const int c = (int)(i & 7);
    const int64_t q = (i >> 3) % S, bh = (i >> 3) / S;
    const int head = (int)(bh % H);
    const int64_t b = bh / H;
    const bf16_t* p = part + (bh * nslab * (int64_t)S + q) * D + 4 * c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // the partials are read exactly once: non-temporal loads, eight slabs requested before the first is added (same slab order)
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    int s = 0;
    for (; s + 8 <= nslab; s += 8) {
        u32x2v v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(p + (int64_t)(s + j) * S * D));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc.x += __uint_as_float(v[j].x << 16); acc.y += __uint_as_float(v[j].x & 0xffff0000u);
            acc.z += __uint_as_float(v[j].y << 16); acc.w += __uint_as_float(v[j].y & 0xffff0000u);
        }
    }
    for (; s < nslab; ++s) {
        const uint2 v = *reinterpret_cast<const uint2*>(p + (int64_t)s * S * D);
        acc.x += __uint_as_float(v.x << 16); acc.y += __uint_as_float(v.x & 0xffff0000u);
        acc.z += __uint_as_float(v.y << 16); acc.w += __uint_as_float(v.y & 0xffff0000u);
    }
    float4 t = make_float4(acc.x * qsc, acc.y * qsc, acc.z * qsc, acc.w * qsc);
    if (freqs) {   // columns 4c .. 4c+3 = two rotation pairs, frequency indices 2c and 2c + 1
        float s0, c0, s1, c1;
        sincosf((float)q * freqs[2 * c], &s0, &c0);
        sincosf((float)q * freqs[2 * c + 1], &s1, &c1);
        t = make_float4(t.x * c0 + t.y * s0, t.y * c0 - t.x * s0, t.z * c1 + t.w * s1, t.w * c1 - t.z * s1);
    }
    *reinterpret_cast<float4*>(dqkv + (b * S + q) * ld + head * D + 4 * c) = t;
}

// combine the key-range parts of the forward: lse = log sum_p exp(lse_p), O = sum_p exp(lse_p - lse) O_p
__global__ void k_attn_combine(const float* __restrict__ o_parts, const float* __restrict__ lse_parts, int P, int64_t o_part,
                               int64_t lse_part, int B, int S, int H, float* __restrict__ o, float* __restrict__ lse) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (row, head, 4-column group)
    const int64_t n = (int64_t)B * S * H * 8;
    if (i >= n) return;
    const int g = (int)(i & 7);
    const int head = (int)((i >> 3) % H);
    const int64_t row = (i >> 3) / H;
    const int64_t li = ((row / S) * H + head) * S + (row % S);
    float mx = -INFINITY;
    for (int p = 0; p < P; ++p) mx = fmaxf(mx, lse_parts[p * lse_part + li]);
    float den = 0.f;
    for (int p = 0; p < P; ++p) den += __expf(lse_parts[p * lse_part + li] - mx);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t oi = row * (H * D) + head * D + 4 * g;
    for (int p = 0; p < P; ++p) {
        const float w = __expf(lse_parts[p * lse_part + li] - mx) / den;
        const float4 v = *reinterpret_cast<const float4*>(o_parts + p * o_part + oi);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    *reinterpret_cast<float4*>(o + oi) = acc;
    if (g == 0) lse[li] = mx + logf(den);
}

// out = sum_p parts[p]  (fixed order)
__global__ void k_sum_parts(const float* __restrict__ parts, int P, int64_t part, int64_t n4, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 acc = reinterpret_cast<const float4*>(parts)[i];
    for (int p = 1; p < P; ++p) {
        const float4 v = reinterpret_cast<const float4*>(parts + p * part)[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
}

// out[row][col0 .. col0 + ncols) = sum_p parts[p][row][col0 ..)  (fixed order; ncols % 4 == 0): the dK / dV columns of the
// query-range parts of the fused backward -- the dq columns of the parts are never written
__global__ void k_sum_cols(const float* __restrict__ parts, int P, int64_t part, int64_t rows, int ld, int col0, int ncols,
                           float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = ncols / 4;
    if (i >= rows * c4) return;
    const int64_t off = (i / c4) * ld + col0 + 4 * (i % c4);
    float4 acc = *reinterpret_cast<const float4*>(parts + off);
    for (int p = 1; p < P; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(parts + p * part + off);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + off) = acc;
}

// number of range parts for a launch whose unsplit grid has `wgs` workgroups: aim at eight workgroups per CU,
// keep at least 1024 streamed rows per part
int split_parts(int64_t wgs, int S) {
    constexpr int tgt = 2048;   // measured at S = 16384 with 1 / 2 / 4 heads: 2048 workgroups beat 512 and 1024;
    if (wgs >= tgt / 2) return 1;   // with 8 heads (1024 workgroups) the unsplit launch ties the split one + its combine pass
    int p = (int)std::min<int64_t>(8, ceil_div(tgt, wgs));
    p = std::min(p, std::max(1, S / 1024));
    return std::max(p, 1);
}
int split_chunk(int S, int P) { return (int)(ceil_div(ceil_div(S, P), 128) * 128); }

}  // namespace

// fused-buffer bf16 path: qkv is ONE fp32 [B*S][(H+2*HKV)*32] projection output
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// bf16 image | 64 B | max |k|^2 per (batch, kv head) (k_key_norm_max)
static size_t key_norm_off(int B, int S, int H, int HKV) { return (sizeof(bf16_t) * (size_t)B * S * (H + 2 * HKV) * D + 64 + 15) & ~(size_t)15; }
static size_t fwd_flags(int B, int S, int H) { return (size_t)B * 8 * (size_t)ceil_div(S, 128) * H; }   // grid of the forward, <= 8 key ranges
static size_t image_only_bytes(int B, int S, int H, int HKV) {
    return align256(key_norm_off(B, S, H, HKV) + sizeof(float) * (size_t)B * KN_BLOCKS * HKV + sizeof(int) * fwd_flags(B, S, H));
}

extern "C" size_t gaot_attn_bf16_image_bytes(int B, int S, int H, int HKV) {
    const int P = split_parts(ceil_div(S, 128) * H * B, S);
    size_t parts = P > 1 ? (size_t)P * ((size_t)B * S * H * D + (size_t)B * H * S) * sizeof(float) : 0;
    return image_only_bytes(B, S, H, HKV) + parts;
}
// the fused backward (k_attn_bwd_fused) is taken when its grid -- one workgroup per 512 keys and kv head -- covers at least
// half of the chip; smaller launches (few heads per rank of a sharded step) keep the two range-split passes
// ... and when its dQ slab partials -- B * H * ceil(S/512) * S * 32 bf16, QUADRATIC in S: 268 MB at S = 16 384, 4.3 GB at
// S = 65 536 -- stay under 1 GiB; longer sequences keep the two-pass form, whose scratch is O(S)
static size_t fused_dqpart_bytes(int B, int S, int H) { return (size_t)B * H * (size_t)ceil_div(S, 512) * (size_t)S * D * sizeof(bf16_t); }
// query-range parts of the fused backward: the smallest power of two that brings its grid to one workgroup per CU, at most 8,
// at least 1024 queries per part
static int fused_parts(int B, int S, int HKV) {
    const int64_t wgs = (int64_t)ceil_div(S, 512) * HKV * B;
    int p = 1;
    while (wgs * p < 256 && p < 8 && S / (2 * p) >= 1024) p *= 2;
    return p;
}
static int fused_chunk(int S, int P) { return (int)(ceil_div(ceil_div(S, P), 64) * 64); }
static bool fused_bwd_ok(int B, int S, int H, int HKV) {
    // (the kernel addresses one batch entry's rows through 31-bit buffer offsets)
    return (int64_t)ceil_div(S, 512) * HKV * B * fused_parts(B, S, HKV) >= 128 && fused_dqpart_bytes(B, S, H) <= ((size_t)1 << 30) &&
           (int64_t)S * (H + 2 * HKV) * D * 2 < 0x7fffffff;
}
extern "C" int gaot_attn_bwd_bf16_fused_eligible(int B, int S, int H, int HKV) { return fused_bwd_ok(B, S, H, HKV) ? 1 : 0; }
static size_t bwd_parts_bytes(int B, int S, int H, int HKV) {   // range parts of the two-pass kernels or of the fused one
    const int P = std::max(split_parts(ceil_div(S, 128) * H * B, S), fused_bwd_ok(B, S, H, HKV) ? fused_parts(B, S, HKV) : 1);
    return P > 1 ? align256((size_t)P * (size_t)B * S * (H + 2 * HKV) * D * sizeof(float)) : 0;
}
extern "C" size_t gaot_attn_bwd_bf16_scratch_bytes(int B, int S, int H, int HKV) {
    // dO image | range-split partial gradients (small grids) | dQ slab partials of the fused pass [B][H][S/512][S][32] fp32
    const size_t dqpart = fused_bwd_ok(B, S, H, HKV) ? fused_dqpart_bytes(B, S, H) : 0;
    return align256(sizeof(bf16_t) * (size_t)B * S * H * D + 64) + bwd_parts_bytes(B, S, H, HKV) + dqpart;
}

extern "C" int gaot_attn_fwd_bf16(const float* qkv, const float* rope_freqs, void* qkv_image, float* o, float* lse,
                                  int B, int S, int H, int HKV, int head_dim, float scale, float dropout_p,
                                  const unsigned long long* dropout_seed, int head0, int heads_total, gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_fwd_bf16: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(qkv_image && o && lse, "null pointer");   // qkv == NULL: the image is already there (gaot_qkv_image)
    GAOT_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)qkv_image | (uintptr_t)o) & 15) == 0, "buffers must be 16-byte aligned");
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seed), "dropout_p in [0,1) and a seed");
    GAOT_CHECK_ARG(heads_total == 0 || (head0 >= 0 && head0 + H <= heads_total), "head0 + H <= heads_total");
    hipStream_t st = (hipStream_t)stream;
    const int ld = (H + 2 * HKV) * D;
    const int64_t rows = (int64_t)B * S;
    const int64_t n = rows * (ld / 2);
    if (qkv)
        GAOT_KLAUNCH(k_prep_qkv, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, qkv, (bf16_t*)qkv_image, rows, ld, H,
                     HKV, S, rope_freqs, scale * LOG2E);
    // bound of the scores of a query row: |q| max_s |k_s| (see the kernel)
    float* kmax2 = reinterpret_cast<float*>(reinterpret_cast<char*>(qkv_image) + key_norm_off(B, S, H, HKV));
    int* redo = reinterpret_cast<int*>(kmax2 + (size_t)B * KN_BLOCKS * HKV);   // per-workgroup flags of the bound-based kernel
    GAOT_KLAUNCH(k_key_norm_max, dim3(KN_BLOCKS, (unsigned)B, (unsigned)HKV), dim3(256), 0, st, (const bf16_t*)qkv_image,
                 (int64_t)ld, S, H, HKV, kmax2);
    // few heads (head-parallel ranks): split the key range over blockIdx.y so that the launch still fills the chip;
    // every part writes a normalised O / lse of its keys into the scratch behind the image, combined below
    const int P = split_parts(ceil_div(S, 128) * H * B, S);
    float* o_parts = reinterpret_cast<float*>(reinterpret_cast<char*>(qkv_image) + image_only_bytes(B, S, H, HKV));
    const int64_t o_part = rows * H * D, lse_part = (int64_t)B * H * S;
    float* lse_parts = o_parts + (size_t)P * o_part;
    FwdArgs a{(const bf16_t*)qkv_image, P > 1 ? o_parts : o, P > 1 ? lse_parts : lse, ld, B, S, H, HKV,
              gdrop::make_drop(dropout_seed, dropout_p, H, head0, heads_total), P > 1 ? split_chunk(S, P) : S, o_part, lse_part, kmax2, redo};
    const dim3 fgrid((unsigned)(ceil_div(S, 128) * H), (unsigned)(P > 1 ? ceil_div(S, a.chunk) : 1), (unsigned)B);
    // (two query blocks per wave, the layout that pays for dK/dV and dQ, gains only 3 % here: the forward is bound by
    // its exp / sum / mask VALU work and loses more from the halved occupancy; a rolled 5-waves/SIMD variant was slower too)
    // with the maximum-free tile path the kernels need ~150 registers: three waves per SIMD without scratch beat four with it
    // (dropout, S = 16 384, 8 heads: <3,4> 0.550 ms, <4,4> 0.66 ms with 144 B of scratch; before that path <4,4> 0.577 ms)
    // bound-based kernel (128 registers: 4 workgroups per CU), then the adaptive one for the workgroups it flagged
    // (schedule pins inside the forward's tile -- start of tile / after the exp stream / before the P V products -- were
    // measured: +-0.5 %, profiles/archive/r4_g_attn_lab.txt; the template parameter stays at 0)
    // (a keep-bit image -- the forward also writing one 32-bit word of keep bits per (query, 32-key tile) for the backward to read
    // through scalar loads -- measured as its two halves, profiles/archive/r4_w_keep_bit_image_lab.txt: WRITING the words (8 and-or steps,
    // one cross-half combine, one 128-byte store per wave and tile) costs the forward 0.516 -> 0.599 ms; the backward with its masks
    // for FREE (no xor, no compares, selects kept) gains 0.834 -> 0.810 ms at most.  Not built.)
    // (K / V stages by LDS-DMA into two stage buffers -- no register staging, no ds_write, ONE barrier per 128 keys, 103 instead
    // of 112 registers -- measured 0.538 / 0.431 ms against 0.524 / 0.424 with / without dropout: the staging is not what the
    // forward waits on; profiles/archive/r4_t_attn_fwd_dma_lab.txt; removed)
    // row sums of the bound-based tile: with dropout one v_dot2c_f32_bf16 per pair of the packed p (-2 %: 0.524 -> 0.514 ms per layer,
    // profiles/archive/r5_l_attn_fwd_rowsum_lab.txt), without dropout the 16 fp32 adds (no difference measured there).
    // GAOT_ATTN_FWD_LAB = 1 / 8 forces the adds / the dot products for both (measurement only).
    static const int fwd_lab = [] { const char* e = getenv("GAOT_ATTN_FWD_LAB"); return e ? atoi(e) : 0; }();
    const bool dot2 = fwd_lab == 8 || (fwd_lab != 1 && a.drop.thr);
    // launches with S a multiple of 512 and at least half a workgroup per CU (key-range parts included): the one-wave-per-SIMD kernel
    // with the generated tile loop (k_attn_fwd_asm) does the bound-based pass; GAOT_ATTN_FWD_ASM=0 keeps the compiled kernel (measurement)
    const bool fwd_asm = [] { const char* e = getenv("GAOT_ATTN_FWD_ASM"); return !e || atoi(e) != 0; }();   // read per call: tests switch it
    if (fwd_asm && fwd_lab == 0 && S % FwdAsmLds::QUERIES == 0 && (P == 1 || a.chunk % FwdAsmLds::KEYS == 0) &&
        (int64_t)(S / FwdAsmLds::QUERIES) * H * B * fgrid.y >= 128) {
        const dim3 agrid((unsigned)((S / FwdAsmLds::QUERIES) * H), fgrid.y, (unsigned)B);
        static const int asm_lab = [] { const char* e = getenv("GAOT_ATTN_FWD_ASM_LAB"); return e ? atoi(e) : 0; }();
        if (asm_lab == 1) {
            if (a.drop.thr) GAOT_KLAUNCH((k_attn_fwd_asm<true, 1>), agrid, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_fwd_asm<false, 1>), agrid, dim3(256), 0, st, a);
        } else if (a.drop.thr) {
            GAOT_KLAUNCH((k_attn_fwd_asm<true>), agrid, dim3(256), 0, st, a);
            GAOT_KLAUNCH((k_attn_fwd_bf16<3, 4, true, false>), fgrid, dim3(256), 0, st, a);
        } else {
            GAOT_KLAUNCH((k_attn_fwd_asm<false>), agrid, dim3(256), 0, st, a);
            GAOT_KLAUNCH((k_attn_fwd_bf16<3, 4, false, false>), fgrid, dim3(256), 0, st, a);
        }
    } else if (a.drop.thr) {
        if (dot2) GAOT_KLAUNCH((k_attn_fwd_bf16<4, 4, true, true, 8>), fgrid, dim3(256), 0, st, a);
        else GAOT_KLAUNCH((k_attn_fwd_bf16<4, 4, true, true>), fgrid, dim3(256), 0, st, a);
        GAOT_KLAUNCH((k_attn_fwd_bf16<3, 4, true, false>), fgrid, dim3(256), 0, st, a);
    } else {
        if (dot2) GAOT_KLAUNCH((k_attn_fwd_bf16<4, 4, false, true, 8>), fgrid, dim3(256), 0, st, a);
        else GAOT_KLAUNCH((k_attn_fwd_bf16<4, 4, false, true>), fgrid, dim3(256), 0, st, a);
        GAOT_KLAUNCH((k_attn_fwd_bf16<3, 4, false, false>), fgrid, dim3(256), 0, st, a);
    }
    if (P > 1)
        GAOT_KLAUNCH(k_attn_combine, dim3((unsigned)ceil_div(rows * H * 8, 256)), dim3(256), 0, st, o_parts, lse_parts,
                           (int)fgrid.y, o_part, lse_part, B, S, H, o, lse);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_attn_bwd_bf16(const void* qkv_image, const float* o, const float* d_o, const float* lse,
                                  void* do_image, float* delta, float* dqkv, const float* rope_freqs, int B, int S, int H,
                                  int HKV, int head_dim, float scale, float dropout_p,
                                  const unsigned long long* dropout_seed, int head0, int heads_total, int phase_mask,
                                  gaot_stream_t stream) {
    GAOT_ENTER();
    if (head_dim != D) {
        gaot_set_error("gaot_attn_bwd_bf16: head_dim %d unsupported (only 32)", head_dim);
        return GAOT_ERR_UNSUPPORTED;
    }
    GAOT_CHECK_ARG(B > 0 && S > 0 && H > 0 && HKV > 0 && H % HKV == 0, "bad shape");
    GAOT_CHECK_ARG(qkv_image && o && (d_o || !(phase_mask & 1)) && lse && do_image && delta && dqkv, "null pointer");
    GAOT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seed), "dropout_p in [0,1) and a seed");
    GAOT_CHECK_ARG(heads_total == 0 || (head0 >= 0 && head0 + H <= heads_total), "head0 + H <= heads_total");
    hipStream_t st = (hipStream_t)stream;
    const int ld = (H + 2 * HKV) * D;
    // range split for small grids (see gaot_attn_fwd_bf16): part p of dK/dV covers queries, of dQ keys
    // [p*chunk, (p+1)*chunk) and accumulates into its own copy of dqkv behind the dO image; summed after phase 4
    const int P0 = split_parts(ceil_div(S, 128) * H * B, S);
    const bool kb_dkv = (int64_t)ceil_div(S, 256) * HKV * B >= 512, kb_dq = (int64_t)ceil_div(S, 256) * H * B >= 512;
    const int P = (kb_dkv || kb_dq) ? 1 : P0;
    const int chunk = P > 1 ? split_chunk(S, P) : S;
    const unsigned ny = (unsigned)(P > 1 ? ceil_div(S, chunk) : 1);
    const int64_t dqkv_part = (int64_t)B * S * ld;
    float* parts = reinterpret_cast<float*>(reinterpret_cast<char*>(do_image) + align256(sizeof(bf16_t) * (size_t)B * S * H * D + 64));
    BwdArgs a{(const bf16_t*)qkv_image, (const bf16_t*)do_image, lse, delta, P > 1 ? parts : dqkv, ld, B, S, H, HKV, scale,
              gdrop::make_drop(dropout_seed, dropout_p, H, head0, heads_total), rope_freqs, chunk, dqkv_part};
    const bool drop = a.drop.thr != 0;
    const int64_t n = (int64_t)B * S * H;
    if (phase_mask & 8)        // do_image already holds the bf16 dO (sequence-parallel exchange): delta only
        GAOT_KLAUNCH(k_delta_bf16, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const bf16_t*)do_image, o, delta, B, S, H);
    else if (phase_mask & 1)
        GAOT_KLAUNCH(k_prep_do, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, d_o, o, (bf16_t*)do_image, delta, B,
                           S, H);
    if (phase_mask & (16 | 32)) {   // 16: dK / dV and dQ slab partials from one pass over the score tiles; 32: the slab reduction
        if (!fused_bwd_ok(B, S, H, HKV)) {
            gaot_set_error("gaot_attn_bwd_bf16: phases 16 / 32 (fused backward) need >= 128 workgroups (ceil(S/512)*HKV*B times up to 8 query parts of >= 1024 rows) and <= 1 GiB of dQ slab partials; use phases 2 and 4");
            return GAOT_ERR_UNSUPPORTED;
        }
        const int nslab = (int)ceil_div(S, 512);      // 512 keys per workgroup in every variant
        bf16_t* dqpart = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(parts) + bwd_parts_bytes(B, S, H, HKV));
        const int Pf = fused_parts(B, S, HKV), chunk_f = Pf > 1 ? fused_chunk(S, Pf) : S;
        const unsigned nyf = (unsigned)(Pf > 1 ? ceil_div(S, chunk_f) : 1);
        BwdArgs af = a;
        af.dqkv = Pf > 1 ? parts : dqkv;      // dK / dV of a query part go to its own copy, summed below
        af.chunk = chunk_f;
        af.dqkv_part = dqkv_part;
        static const int lab = [] { const char* e = getenv("GAOT_ATTN_BWD_LAB"); return e ? atoi(e) : 0; }();   // lab: schedule pins
        FusedArgs fa{af, dqpart, nslab, lab, nullptr};
        static const bool want_stamps = [] { const char* e = getenv("GAOT_ATTN_BWD_STAMPS"); return e && atoi(e) != 0; }();
        static unsigned long long* stamp_buf = nullptr;
        const size_t stamp_n = (size_t)nslab * HKV * B * nyf * 8 * 16;
        const dim3 gf((unsigned)(nslab * HKV), nyf, (unsigned)B);
        auto go = [&](auto kern, int lds_bytes, int nthr) -> int {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) {
                gaot_set_error("gaot_attn_bwd_bf16: cannot set dynamic LDS %d for the fused backward", lds_bytes);
                return GAOT_ERR_LAUNCH;
            }
            GAOT_KLAUNCH(kern, gf, dim3(nthr), lds_bytes, st, fa);
            return GAOT_OK;
        };
        if (phase_mask & 16) {
            // lab switch (measurement only): GAOT_ATTN_BWD_VARIANT = 0 (the shipped form: 8 waves x 2 key blocks, 64-query stages,
            // packed row words + SDWA compares), 1 (4 waves x 4 key blocks, one wave per SIMD), 7 (round-3 mask arithmetic);
            // GAOT_ATTN_BWD_STAMPS=1: the diagnostic instantiation with in-kernel cycle stamps (results unchanged, slower)
            static const int variant = [] { const char* e = getenv("GAOT_ATTN_BWD_VARIANT"); return e ? atoi(e) : 0; }();
            int rc;
            // k_attn_bwd_asm (one wave per SIMD, hand-scheduled tile loop, dS transposed on the matrix pipe with dropout / through LDS
            // without) is the default for whole-sequence launches: same box, S = 16 384, H = 8 (profiles/archive/r5_d_attn_bwd_asm_mfma_t2.txt):
            // 0.798 against 0.836 ms with dropout (-4.5 %), 0.669 against 0.717 ms without (-6.7 %); dK / dV bit-identical to the compiled
            // kernel, dQ within 5e-5 of peak (four slots of four key blocks instead of eight of two).  Few heads per launch: query-range
            // parts as in the compiled kernel (ranges that are multiples of 128 queries; others keep the compiled kernel).  GAOT_ATTN_BWD_VARIANT=2 forces the asm kernel, =3 the compiled one.
            if ((variant == 2 || (variant == 0 && lab == 0 && !want_stamps)) && (nyf == 1 || chunk_f % AsmLds<true>::QS == 0)) {
                if (lab == 1) rc = drop ? go(k_attn_bwd_asm<true, 1>, AsmLds<true>::TOTAL, 256) : go(k_attn_bwd_asm<false, 1>, AsmLds<false>::TOTAL, 256);
                else if (lab == 2) rc = drop ? go(k_attn_bwd_asm<true, 2>, AsmLds<true>::TOTAL, 256) : go(k_attn_bwd_asm<false, 2>, AsmLds<false>::TOTAL, 256);
                else if (lab == 3) rc = drop ? go(k_attn_bwd_asm<true, 3>, AsmLds<true>::TOTAL, 256) : go(k_attn_bwd_asm<false, 3>, AsmLds<false>::TOTAL, 256);
                else rc = drop ? go(k_attn_bwd_asm<true>, AsmLds<true>::TOTAL, 256) : go(k_attn_bwd_asm<false>, AsmLds<false>::TOTAL, 256);
            }
            else if (variant == 1)       // one wave per SIMD, compiler-managed 512 registers: 1.9 / 1.27 ms (profiles/archive/r4_b_attn_bwd_lab.txt)
                rc = drop ? go(k_attn_bwd_fused<true, 4, 4, 2>, FusedLds<4, 4, 2>::TOTAL, 256) : go(k_attn_bwd_fused<false, 4, 4, 2>, FusedLds<4, 4, 2>::TOTAL, 256);
            // (128-query stages with bf16 slots -- FB_NT = 4 -- measured 1.07 / 0.81 ms against 0.90 / 0.76: spills in the
            // dropout variant, profiles/archive/r4_g_attn_lab.txt; removed)
            else if (variant == 7 && drop)   // round-3 mask arithmetic (row words per query, xor + compare per element)
                rc = go(k_attn_bwd_fused<true, 8, 2, 2, false, 0, 26>, FusedLds<8, 2, 2>::TOTAL, 512);
            else if (want_stamps && drop) {    // diagnostic build: cycles per phase of the PK kernel, printed to stderr
                if (!stamp_buf) (void)hipMalloc((void**)&stamp_buf, stamp_n * 8);
                fa.stamps = stamp_buf;
                rc = go(k_attn_bwd_fused<true, 8, 2, 2, true, 2, 26>, FusedLds<8, 2, 2>::TOTAL, 512);
                if (rc == GAOT_OK && stamp_buf) {
                    (void)hipStreamSynchronize(st);
                    unsigned long long* h = (unsigned long long*)malloc(stamp_n * 8);
                    (void)hipMemcpy(h, stamp_buf, stamp_n * 8, hipMemcpyDeviceToHost);
                    double sum[16] = {0}, lo[16] = {0}, hi[16] = {0};   // all waves / waves 0-3 / waves 4-7
                    const size_t nw = stamp_n / 16;
                    for (size_t w = 0; w < nw; ++w)
                        for (int i = 0; i < 16; ++i) { sum[i] += (double)h[w * 16 + i]; ((w & 7) < 4 ? lo : hi)[i] += (double)h[w * 16 + i]; }
                    double tot = 0;
                    for (int i = 0; i < 16; ++i) tot += sum[i];
                    fprintf(stderr, "[attn_bwd stamps] mean cycles per wave %.0f;", tot / nw);
                    for (int i = 0; i < 10; ++i) fprintf(stderr, " s%d %.1f%% (w0-3 %.1f%% w4-7 %.1f%%)", i, 100 * sum[i] / tot, 200 * lo[i] / tot, 200 * hi[i] / tot);
                    fprintf(stderr, "\n");
                    free(h);
                }
            }

            else if (lab == 200)      // TT: dS transposed on the matrix pipe
                rc = drop ? go(k_attn_bwd_fused<true, 8, 2, 2, true, 0, 26, true>, FusedLds<8, 2, 2>::TOTAL, 512) : go(k_attn_bwd_fused<false, 8, 2, 2, false, 0, 26, true>, FusedLds<8, 2, 2>::TOTAL, 512);
            else if (lab == 100)      // no schedule pins (the round-3 / r4_d schedule)
                rc = drop ? go(k_attn_bwd_fused<true, 8, 2, 2, true, 0, 0>, FusedLds<8, 2, 2>::TOTAL, 512) : go(k_attn_bwd_fused<false, 8, 2, 2, false, 0, 0>, FusedLds<8, 2, 2>::TOTAL, 512);
            // (compile-time priority variants on the pinned schedule -- waves >= W/2 one level higher, static priority without
            // flips, no priority at all -- measured within +-1.5 % of the flips: profiles/archive/r4_h_attn_bwd_prio_stamps.txt)
            // (double-buffered stage tiles filled by the first half of the waves before the stage barrier -- to use their 17 % of
            // barrier idle time -- measured 0.97 / 0.795 ms against 0.89 / 0.786: profiles/archive/r4_i_attn_bwd_double_buffer_lab.txt; removed)
            // (the mask stream of a register group -- 4 xor, 8 exp, 8 SDWA compares, 16 selects, 8 multiplies -- as ONE asm statement,
            // which removes the s_nop the compiler puts behind every asm boundary (60 per stage): 0.856 -> 0.969 ms, exp-first order or
            // not; the nops are not what the stream waits on -- profiles/archive/r4_o_attn_bwd_asm_group_lab.txt; removed)
            // (all S products before the dP products, with and without a pin behind the cluster, and a pin after the dV / dK block:
            // within +-1 % -- profiles/archive/r4_k_attn_bwd_cluster_order_lab.txt)
            else                      // shipped: schedule pinned at the top of a tile, after the exp / mask stream and after the dQ products
                rc = drop ? go(k_attn_bwd_fused<true, 8, 2, 2, true, 0, 26>, FusedLds<8, 2, 2>::TOTAL, 512) : go(k_attn_bwd_fused<false, 8, 2, 2, false, 0, 26>, FusedLds<8, 2, 2>::TOTAL, 512);
            if (rc != GAOT_OK) return rc;
        }
        if ((phase_mask & 16) && Pf > 1)   // dK / dV columns: sum of the query parts, part order
            GAOT_KLAUNCH(k_sum_cols, dim3((unsigned)ceil_div((int64_t)B * S * (2 * HKV * D / 4), 256)), dim3(256), 0, st, parts, (int)nyf,
                         dqkv_part, (int64_t)B * S, ld, H * D, 2 * HKV * D, dqkv);
        const float qsc = drop ? scale * a.drop.inv_keep : scale;
        if (phase_mask & 32)
            GAOT_KLAUNCH(k_attn_dq_reduce, dim3((unsigned)ceil_div((int64_t)B * H * S * 8, 256)), dim3(256), 0, st, dqpart, nslab, B, S, H,
                     ld, qsc, rope_freqs, dqkv);
    }
    if (phase_mask & 2) {
        // two key blocks per wave halve the LDS reads per (query tile, key block) unit; taken when its grid
        // (256 keys per workgroup) still gives every CU two workgroups
        const dim3 gkb((unsigned)(ceil_div(S, 256) * HKV), 1, (unsigned)B), g1((unsigned)(ceil_div(S, 128) * HKV), ny, (unsigned)B);
        // one key block per wave: 2 workgroups per CU (256 registers) when the grid is small or the dropout words are
        // live (at 4 per CU the dropout variant spills: 0.83 -> 0.49 ms at S = 16384, H = 4)
        const bool occ2 = drop || (int64_t)g1.x * ny * B <= 512;
        if (kb_dkv) {
            // dropout: 64-query stages (NT = 2) keep the kernel free of scratch spills (237 registers; the 128-query
            // form spilled 25) and measure 3 % faster: 0.848 -> 0.823 ms at S = 16384, H = 8
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dkv_kb<2, 2, 2, true>), gkb, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dkv_kb<2, 4, 2, false>), gkb, dim3(256), 0, st, a);
        } else if (occ2) {
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dkv_bf16<2, true>), g1, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dkv_bf16<2, false>), g1, dim3(256), 0, st, a);
        } else {
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dkv_bf16<4, true>), g1, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dkv_bf16<4, false>), g1, dim3(256), 0, st, a);
        }
    }
    if (phase_mask & 4) {
        const dim3 gkb((unsigned)(ceil_div(S, 256) * H), 1, (unsigned)B), g1((unsigned)(ceil_div(S, 128) * H), ny, (unsigned)B);
        const bool occ2 = drop || (int64_t)g1.x * ny * B <= 512;
        if (kb_dq) {   // two query blocks per wave: 0.50 -> 0.43 ms at S = 16384, H = 8
            // dropout: 64-key stages, no scratch spills (the 128-key form spilled 11 registers), same time
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dq_kb<2, 2, 2, true>), gkb, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dq_kb<2, 4, 2, false>), gkb, dim3(256), 0, st, a);
        } else if (occ2) {
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dq_bf16<2, true>), g1, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dq_bf16<2, false>), g1, dim3(256), 0, st, a);
        } else {
            if (drop) GAOT_KLAUNCH((k_attn_bwd_dq_bf16<4, true>), g1, dim3(256), 0, st, a);
            else GAOT_KLAUNCH((k_attn_bwd_dq_bf16<4, false>), g1, dim3(256), 0, st, a);
        }
        if (P > 1)   // dK/dV (phase 2) and dQ parts are complete: fixed-order sum into dqkv
            GAOT_KLAUNCH(k_sum_parts, dim3((unsigned)ceil_div(dqkv_part / 4, 256)), dim3(256), 0, st, parts, (int)ny, dqkv_part,
                               dqkv_part / 4, dqkv);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
