// The SwiGLU FFN of a latent Transformer block as ONE kernel per direction over 64-row blocks (reference attn.py:146-157:
// w2(silu(w1 x) * w3 x), and :226-229 for the residual that follows it), d_model = 256, bf16 operands, fp32 accumulation.
//
// Why a row-block kernel and what it is bound by.  Unfused, the forward is two launches (w1|w3 + SwiGLU epilogue, then w2 + residual)
// that move the [S, F] product u through HBM in between and each pay their own fill / drain; the backward is four (cast of dy, du = dy W2,
// SwiGLU', dx = dag W13) that move du ([S, F]) and re-read a | g.  A workgroup here owns 64 rows and walks F in chunks of 128 columns:
//     forward :  a|g chunk = h W13c^T (K = 256)  ->  u chunk = silu(a) g (registers -> LDS, 16 KB)  ->  y += u chunk W2c^T (K = 128)
// so u never returns from HBM.  What a row block cannot share with its neighbours is the WEIGHTS: every workgroup streams all of W13
// and W2 (1.5 MB of bf16 at F = 1024) from its XCD's L2.  The operands of a wave are its OWN 64 weight rows of every step, nothing a
// second wave reads, so they skip LDS: the weights are PRE-PACKED IN MFMA FRAGMENT ORDER (k_ffn_pack: one 1-KB block per
// wave-instruction, lane l's 16 bytes at block + 16 l) and loaded global -> VGPR with fully coalesced 1-KB requests, two steps ahead
// of their use (register ring of three).  LDS holds only the activation tile (32 KB, LDS-DMA, source-side XOR swizzle as in
// gemm_k256.hip) and two u chunks.  Arithmetic and summation order are those of the unfused kernels (k_gemm_k256<OUT_SWIGLU>,
// k_gemm_tn_n256): a | g, u and y are BIT-IDENTICAL to the two-launch path, so the backward and every test bound are unchanged.
// Per 64-row block and chunk a wave issues 96 MFMAs (3 072 matrix-pipe cycles); HBM: a | g and u are still written once for the
// backward (96 MB per layer at S = 16 384), which is this kernel's floor (~16 us at the measured copy rate) -- see DESIGN.md §3.4.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// lab only (tools/lab/ffn_fwd_lab.hip): ablation bits -- 1: the weight ring is loaded once and never refilled; 4: u = a (no SwiGLU
// arithmetic); 8: the activation / u fragments are read from LDS once per chunk phase; 16: no barrier in the chunk loop
#ifndef GAOT_FFN_ABL
#define GAOT_FFN_ABL 0
#endif
#ifdef GAOT_FFN_TIMING      // lab only: s_memtime stamps of workgroup 0 / wave 0 at the phase boundaries
__device__ unsigned long long g_ffn_t[64];
#define FFN_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ffn_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FFN_STAMP(i) do { } while (0)
#endif
constexpr int D = 256, RB = 64, FC = 128;          // d_model, rows per workgroup, F columns per chunk
constexpr int H_BYTES = RB * D * 2;                // activation tile: 64 rows x 512 B
constexpr int U_BYTES = RB * FC * 2;               // one u chunk: 64 rows x 256 B
constexpr int AG_BYTES = RB * 2 * FC * 2;          // one a | g chunk on its way to HBM: 64 rows x 512 B (SAVE only)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(bf16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)b) << 16);
}
__device__ __forceinline__ float rbf(float v) { return __uint_as_float((unsigned)__builtin_bit_cast(bf16_t, (__bf16)v) << 16); }

// ---- weight packing ---------------------------------------------------------------------------------------------------------------
// Fragment block = what one wave-instruction loads: 64 lanes x 8 bf16; lane (l31, hf) holds W[row0 + l31][k0 + 8 hf + 0..7] -- the A
// operand of v_mfma_f32_32x32x16_bf16 in the transposed product (weight rows on the tile rows).
//   forward  W13p: block (((c*4 + w)*2 + jt)*16 + s): row = jt*F + c*128 + w*32 + l31 of [w1; w3] ([2F][256]), k = 16 s + 8 hf + e
//            W2p : block (((c*4 + w)*2 + jt)* 8 + s): row = w*64 + jt*32 + l31 of w2 ([256][F]),         k = c*128 + 16 s + 8 hf + e
//   backward W2tp: du = dy W2, fragment rows = F columns: block (((c*4 + w)*16 + s): "row" f = c*128 + w*32 + l31, k = 16 s + 8 hf + e
//                  of W2^T, i.e. element w2[k][f]
//            W13tp: dx = dag W13, fragment rows = d columns: block ((((c*4 + w)*2 + jt)*16 + s): "row" j = w*64 + jt*32 + l31,
//                  k-index kk = 16 s + 8 hf + e over the chunk's 256 dag columns (kk < 128: a column c*128 + kk, else g column
//                  c*128 + kk - 128), i.e. element w13[(kk < 128 ? 0 : F) + c*128 + (kk & 127)][j]
__device__ __forceinline__ void ffn_pack_body(const float* __restrict__ w13, const float* __restrict__ w2, int F, bf16_t* __restrict__ w13p,
                                              bf16_t* __restrict__ w2p, bf16_t* __restrict__ w2tp, bf16_t* __restrict__ w13tp) {
    const int NC = F / FC;
    const int64_t n13 = (int64_t)NC * 4 * 2 * 16 * 64, n2 = (int64_t)NC * 4 * 2 * 8 * 64, n2t = (int64_t)NC * 4 * 16 * 64;
    const int64_t total = n13 + n2 + (w2tp ? n2t + n13 : 0);
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = id;
        const float* src;
        int64_t stride;      // element stride between consecutive e
        bf16_t* dst;
        if (t < n13) {
            const int lane = t & 63, s = (t >> 6) & 15, jt = (t >> 10) & 1, w = (t >> 11) & 3, c = (int)(t >> 13);
            const int row = jt * F + c * FC + w * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5);
            src = w13 + (int64_t)row * D + k; stride = 1; dst = w13p + t * 8;
        } else if ((t -= n13) < n2) {
            const int lane = t & 63, s = (t >> 6) & 7, jt = (t >> 9) & 1, w = (t >> 10) & 3, c = (int)(t >> 12);
            const int row = w * 64 + jt * 32 + (lane & 31), k = c * FC + 16 * s + 8 * (lane >> 5);
            src = w2 + (int64_t)row * F + k; stride = 1; dst = w2p + t * 8;
        } else if ((t -= n2) < n2t) {
            const int lane = t & 63, s = (t >> 6) & 15, w = (t >> 10) & 3, c = (int)(t >> 12);
            const int f = c * FC + w * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5);
            src = w2 + (int64_t)k * F + f; stride = F; dst = w2tp + t * 8;
        } else {
            t -= n2t;
            const int lane = t & 63, s = (t >> 6) & 15, jt = (t >> 10) & 1, w = (t >> 11) & 3, c = (int)(t >> 13);
            const int j = w * 64 + jt * 32 + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
            const int row = (kk < FC ? 0 : F) + c * FC + (kk & (FC - 1));
            src = w13 + (int64_t)row * D + j; stride = D; dst = w13tp + t * 8;
        }
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2(src[(2 * e) * stride], src[(2 * e + 1) * stride]);
        *reinterpret_cast<u32x4*>(dst) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

constexpr int PACK_MAX = 16;
struct PackTable {
    const float* w13[PACK_MAX];
    const float* w2[PACK_MAX];
    const float* wo[PACK_MAX];      // o_proj.weight [256][256] or null: a fifth image behind the four (gaot_block_pack_multi)
    bf16_t* packed[PACK_MAX];
};
// blockIdx.y = the FFN: all blocks of a Transformer in one launch
__global__ void k_ffn_pack(PackTable t, int F, int with_backward) {
    bf16_t* p = t.packed[blockIdx.y];
    bf16_t* w2p = p + (int64_t)2 * F * D;
    bf16_t* w2tp = with_backward ? w2p + (int64_t)D * F : nullptr;
    ffn_pack_body(t.w13[blockIdx.y], t.w2[blockIdx.y], F, p, w2p, w2tp, with_backward ? w2tp + (int64_t)D * F : nullptr);
    if (const float* wo = t.wo[blockIdx.y]) {
        // block ((w*2 + jt)*16 + s): row 64 w + 32 jt + l31 of o_proj.weight, k = 16 s + 8 hf + e
        bf16_t* wop = p + ((int64_t)2 * F * D + (int64_t)D * F) * (with_backward ? 2 : 1);
        for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < D * D / 8; id += gridDim.x * blockDim.x) {
            const int lane = id & 63, s2 = (id >> 6) & 15, jt = (id >> 10) & 1, w = id >> 11;
            const float* src = wo + (int64_t)(w * 64 + jt * 32 + (lane & 31)) * D + 16 * s2 + 8 * (lane >> 5);
            *reinterpret_cast<u32x4*>(wop + (int64_t)id * 8) = u32x4{pack2(src[0], src[1]), pack2(src[2], src[3]), pack2(src[4], src[5]), pack2(src[6], src[7])};
            // the same blocks of o_proj.weight^T (the backward's d_o = dh Wo: "row" j = 64 w + 32 jt + l31, k = 16 s + 8 hf + e -> wo[k][j])
            const float* st = wo + (int64_t)(16 * s2 + 8 * (lane >> 5)) * D + (w * 64 + jt * 32 + (lane & 31));
            *reinterpret_cast<u32x4*>(wop + (int64_t)D * D + (int64_t)id * 8) =
                u32x4{pack2(st[0], st[D]), pack2(st[2 * D], st[3 * D]), pack2(st[4 * D], st[5 * D]), pack2(st[6 * D], st[7 * D])};
        }
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------------------------
// X [M][256] bf16 (the normalised h RMSNorm wrote), R [M][ldr] fp32 residual or null, Y [M][256] fp32, AG [M][2F] bf16 (a | g),
// U [M][F] bf16.  SAVE = false: a | g and u are not written (inference / a backward that recomputes them).
// NORM (gaot_norm_ffn_fwd): the block's RMSNorm in front of the FFN (reference attn.py:227-229: h = ffn_norm(h); h + ffn(h)) computed
// HERE from the fp32 rows -- X is not read; R = the un-normalised rows [M][ldr], NW the norm weight: a wave normalises its 16 rows with
// the arithmetic of k_rmsnorm_fwd (one wave per row, a lane per float4, the same butterfly), writes them to the LDS tile and to YB (bf16
// [M][256]: the backward's input and the dW13 product's operand) and their 1/rms to RSTD; the residual added at the end is the
// normalised row, recomputed from R.  The stand-alone norm pass (a read and two writes of [M][256]) is gone.
// OPROJ (gaot_block_tail_fwd; with NORM): the attention's output projection and the block's first residual in front of that norm
// (reference attn.py:127, 226: h = x + o_proj(attn)): O = the attention output fp32 [M][ldo], WOp = o_proj.weight as fragment blocks
// ((w*2 + jt)*16 + s: row 64 w + 32 jt + l31, k = 16 s + 8 hf + e), R = x; the product's accumulators start as x, h is written to H
// (fp32 [M][256]: the backward's norm input) and normalised from the registers -- the sum of squares of a row is the sum of the four
// waves' 64-column partials (another order than k_rmsnorm_fwd: h itself is bit-identical to the stand-alone GEMM, 1/rms to rounding).
struct TailArgs {
    const float* NW; float eps; bf16_t* YB; float* RSTD;      // NORM
    const float* O; int ldo; const u32x4* WOp; float* H;      // OPROJ
};
template <bool SAVE, int RD, bool NORM = false, bool OPROJ = false>
__global__ __launch_bounds__(256, 1) void k_ffn_fwd(const bf16_t* __restrict__ X, const u32x4* __restrict__ W13p,
                                                     const u32x4* __restrict__ W2p, const float* __restrict__ R, float* __restrict__ Y,
                                                     bf16_t* __restrict__ AG, bf16_t* __restrict__ U, int M, int F, int ldr, TailArgs ta) {
    static_assert(!OPROJ || NORM, "the output projection comes with the norm");
    const float* __restrict__ NW = ta.NW;
    const float eps = ta.eps;
    bf16_t* __restrict__ YB = ta.YB;
    float* __restrict__ RSTD = ta.RSTD;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    // consecutive workgroup ids sit on consecutive XCDs: give every XCD a contiguous range of row blocks (the a | g / u / y rows one
    // XCD's L2 write-combines are then neighbours)
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB, NC = F / FC;
    FFN_STAMP(0);
    constexpr int LA = RD - 1;
    static_assert(6 % RD == 0, "the ring is indexed by the position within the loop body");

    const int64_t xbytes = (int64_t)M * D * 2, agbytes = (int64_t)M * F * 4, ubytes = (int64_t)M * F * 2, ybytes = (int64_t)M * D * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t agrs = __builtin_amdgcn_make_buffer_rsrc((void*)(SAVE ? AG : nullptr), 0, SAVE ? (int)(agbytes > 0x7fffffff ? 0x7fffffff : agbytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)(SAVE ? U : nullptr), 0, SAVE ? (int)(ubytes > 0x7fffffff ? 0x7fffffff : ubytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)(ybytes > 0x7fffffff ? 0x7fffffff : ybytes), 0x00020000);

    // the residual rows of this lane's outputs: requested at the top of the LAST chunk's iteration (below) -- at the kernel's end the
    // workgroup would finish on an exposed HBM round trip, at its start all 256 workgroups would wait for 16 MB more before their first MFMA
    f32x4 rres[2][2][4];
    const float* rsrc_p = OPROJ ? ta.H : R;          // the rows the final residual is (re)computed from
    const int ldres = OPROJ ? D : ldr;
    const int64_t rbytes = (int64_t)M * ldres * 4;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)rsrc_p, 0, rsrc_p ? (int)(rbytes > 0x7fffffff ? 0x7fffffff : rbytes) : 0, 0x00020000);
    // A STEP = 8 fragment blocks (8 KB per wave) and 16 MFMAs.  Steps of chunk c: st 0..3 = the four 64-deep k-slices of h W13c^T (tiles
    // a, g), st 4, 5 = the two 64-deep k-slices of u W2c^T (tiles: output columns 64 w .. +31, +32 .. +63).  Chunks past the last one read
    // as zeros (buffer range check): the loop below runs one a | g product too many instead of carrying a second copy of its body.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t w13rs = __builtin_amdgcn_make_buffer_rsrc((void*)W13p, 0, 2 * F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)W2p, 0, D * F * 2, 0x00020000);
    bool wfirst = true;
    auto wload = [&](u32x4 (&dst)[8], int c, int st) {
        if ((GAOT_FFN_ABL & 1) && !wfirst) return;
        if (st < 4) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    dst[jt * 4 + s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w13rs, lane * 16, ((((c * 4 + wv) * 2 + jt) * 16) + 4 * st + s) * 1024, 0));
        } else {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    dst[jt * 4 + s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16, ((((c * 4 + wv) * 2 + jt) * 8) + 4 * (st - 4) + s) * 1024, 0));
        }
    };
    // The wave's timeline of steps: positions 0..3 = a | g product of chunk 0 (prologue); then per loop iteration c six positions:
    // b = 0..3 the a | g product of chunk c + 1, b = 4, 5 the y accumulation of chunk c.  Ring slot of a position = position % RD
    // (6 % RD == 0: static); LA = RD - 1 steps (8 KB per wave each) are in flight.  The XCD's L2 hands a CU ~75 GB/s when all 32
    // stream (16 channels x 64 B/clk): the 1.5 MB of weights per row block cost ~20 us of that pipe.
    u32x4 wr[RD][8];
    auto prefetch_body = [&](int c, int tpos) {     // tpos = position in the body of iteration c, may run into iteration c + 1
        const int cc = tpos < 6 ? c : c + 1, b = tpos < 6 ? tpos : tpos - 6;
        wload(wr[(4 + tpos) % RD], b < 4 ? cc + 1 : cc, b);
    };
#pragma unroll
    for (int p = 0; p < LA; ++p) {
        if (p < 4) wload(wr[p % RD], 0, p);
        else prefetch_body(0, p - 4);
    }
    if (GAOT_FFN_ABL & 1) { wload(wr[LA % RD], 0, 0); wfirst = false; }

    float* rstd_l = reinterpret_cast<float*>(lds + H_BYTES + 2 * U_BYTES + (SAVE ? 2 * AG_BYTES : 0));      // NORM: 64 floats (+ OPROJ: 4 x 64)
    if constexpr (OPROJ) {
        const int sw0 = l31 & 15;
        const int64_t obytes = (int64_t)M * ta.ldo * 4, xb2 = (int64_t)M * ldr * 4, hbytes = (int64_t)M * D * 4, ybbytes = (int64_t)M * D * 2;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)ta.O, 0, (int)(obytes > 0x7fffffff ? 0x7fffffff : obytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, R ? (int)(xb2 > 0x7fffffff ? 0x7fffffff : xb2) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)ta.H, 0, (int)(hbytes > 0x7fffffff ? 0x7fffffff : hbytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc((void*)YB, 0, (int)(ybbytes > 0x7fffffff ? 0x7fffffff : ybbytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t wors = __builtin_amdgcn_make_buffer_rsrc((void*)ta.WOp, 0, D * D * 2, 0x00020000);
        // the attention output rows fp32 -> bf16 -> LDS tile (row 8 i + tid >> 5, 16-byte bf16 chunk tid & 31 = 8 floats)
        {
            f32x4 d0[8], d1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
                const unsigned off = m < M ? (unsigned)m * (unsigned)ta.ldo * 4u + (unsigned)(threadIdx.x & 31) * 32u : 0x80000000u;
                d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, off, 0, 0));
                d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, off, 16, 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31;
                const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
                *reinterpret_cast<u32x4*>(lds + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
            }
        }
        // h starts as x (the block's first residual): column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31
        f32x16 hacc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * i + l31;
            const unsigned roff = (m < M && R) ? (unsigned)m * (unsigned)ldr * 4u : 0x80000000u;     // out of range -> zeros
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs2, roff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0));
                    hacc[jt][i][4 * q] = v[0]; hacc[jt][i][4 * q + 1] = v[1]; hacc[jt][i][4 * q + 2] = v[2]; hacc[jt][i][4 * q + 3] = v[3];
                }
        }
        u32x4 wo[2][8];
        auto woload = [&](u32x4 (&dst)[8], int st) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wors, lane * 16, (((wv * 2 + jt) * 16) + 4 * st + s2) * 1024, 0));
        };
        woload(wo[0], 0);
        woload(wo[1], 1);
        __builtin_amdgcn_s_barrier();       // the bf16 tile of the attention output is complete
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int slot = ((2 * (4 * st + s2) + hf) ^ sw0) << 4;
                const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(lds + l31 * 512 + slot), o1 = *reinterpret_cast<const bf16x8*>(lds + (32 + l31) * 512 + slot);
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wo[st & 1][s2]), w1 = __builtin_bit_cast(bf16x8, wo[st & 1][4 + s2]);
                hacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, o0, hacc[0][0], 0, 0, 0);
                hacc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, o0, hacc[1][0], 0, 0, 0);
                hacc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, o1, hacc[0][1], 0, 0, 0);
                hacc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, o1, hacc[1][1], 0, 0, 0);
            }
            if (st + 2 < 4) woload(wo[st & 1], st + 2);
        }
        // h to HBM (the backward's norm input) and the rows' sums of squares: this wave's 64 columns, then the four waves' partials
        float* part = rstd_l + 64;       // [4 waves][64 rows]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * i + l31;
            const unsigned rowoff = m < M ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;
            float ssl = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {hacc[jt][i][4 * q], hacc[jt][i][4 * q + 1], hacc[jt][i][4 * q + 2], hacc[jt][i][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), hrs, rowoff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0);
                    ssl += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                }
            ssl += __shfl_xor(ssl, 32, 64);
            if (hf == 0) part[wave * 64 + 32 * i + l31] = ssl;
        }
        __builtin_amdgcn_s_barrier();       // partials complete; every wave is done reading the attention-output tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = 32 * i + l31, m = m0 + ml;
            const float ssr = ((part[ml] + part[64 + ml]) + part[128 + ml]) + part[192 + ml];
            const float r = rsqrtf(ssr / (float)D + eps);
            if (wave == 0 && hf == 0) {
                rstd_l[ml] = r;
                if (m < M) RSTD[m] = r;
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(NW + wave * 64 + 32 * jt + 8 * q + 4 * hf);
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 pk = {pack2(hacc[jt][i][4 * q] * r * g[0], hacc[jt][i][4 * q + 1] * r * g[1]),
                                      pack2(hacc[jt][i][4 * q + 2] * r * g[2], hacc[jt][i][4 * q + 3] * r * g[3])};
                    // tile: row ml, 16-byte chunk 8 wave + 4 jt + q, its 8-byte half hf
                    *reinterpret_cast<u32x2*>(lds + ml * 512 + (((8 * wave + 4 * jt + q) ^ (ml & 15)) << 4) + 8 * hf) = pk;
                }
        }
        __builtin_amdgcn_s_barrier();       // the normalised tile is complete
        // bf16(norm(h)) to HBM, this wave's 16 rows, two rows per instruction
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = 16 * wave + 2 * k + hf, ch = l31, m = m0 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + row * 512 + ((ch ^ (row & 15)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(v, ybrs, m < M ? (unsigned)m * (unsigned)D * 2u + (unsigned)ch * 16u : 0x80000000u, 0, 0);
        }
    } else if constexpr (NORM) {
        // rows 16 wave .. + 15 of the block, a lane per float4 (the arithmetic and summation order of k_rmsnorm_fwd, rowops.hip)
        const int64_t rbytes = (int64_t)M * ldr * 4, ybbytes = (int64_t)M * D * 2;
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)(rbytes > 0x7fffffff ? 0x7fffffff : rbytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc((void*)YB, 0, (int)(ybbytes > 0x7fffffff ? 0x7fffffff : ybbytes), 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(NW)[lane];
        float4 v[16];
        float ss[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = m0 + wave * 16 + j;
            v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, m < M ? (unsigned)m * (unsigned)ldr * 4u + lane * 16u : 0x80000000u, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            ss[j] = 0.f;
            ss[j] += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 16; ++j) ss[j] += __shfl_xor(ss[j], o, 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rl = wave * 16 + j, m = m0 + rl;
            const float r = rsqrtf(ss[j] / (float)D + eps);
            const float4 o = make_float4(v[j].x * r * g.x, v[j].y * r * g.y, v[j].z * r * g.z, v[j].w * r * g.w);
            const u32x4 dummy = {0, 0, 0, 0};
            (void)dummy;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 pk = {pack2(o.x, o.y), pack2(o.z, o.w)};
            // tile: row rl, 8-byte half (lane & 1) of the 16-byte chunk lane >> 1, chunk stored at slot chunk ^ (rl & 15)
            *reinterpret_cast<u32x2*>(lds + rl * 512 + ((((lane >> 1) ^ (rl & 15))) << 4) + 8 * (lane & 1)) = pk;
            __builtin_amdgcn_raw_buffer_store_b64(pk, ybrs, m < M ? (unsigned)m * (unsigned)D * 2u + lane * 8u : 0x80000000u, 0, 0);
            if (lane == 0) {
                rstd_l[rl] = r;
                if (m < M) RSTD[m] = r;
            }
        }
    } else {
        // the activation tile by LDS-DMA: piece q = wave*8 + i fills tile rows 2q, 2q+1; slot s of row r holds source chunk s ^ (r & 15)
        const int lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = 2 * i + lh;
            const int voff = (m0 + wave * 16 + x) * D * 2 + (((lane & 31) ^ x) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + i) * 1024), 16, voff, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA pieces have landed (the NORM forms write the tile with LDS stores: a
                                                                // drain here would sit out the round trip of their h / yb stores to HBM)
    }
    __builtin_amdgcn_s_barrier();
    FFN_STAMP(1);

    f32x16 y[2][2], agc[2][2], agn[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { y[jt][i][r] = 0.f; agc[jt][i][r] = 0.f; }

    const int sw = l31 & 15;
    const char* hb = lds + l31 * 512;                       // activation rows l31 (i = 0) and 32 + l31 (+ 32 * 512)
    bf16x8 hn0, hn1;
    // one 16-deep k-step (ks = 0..15) of the a | g product into acc, with the NEXT k-step's activation fragments requested first
    auto ag_kstep = [&](f32x16 (&acc)[2][2], const u32x4 (&w)[8], int ks) {
        const bf16x8 h0 = hn0, h1 = hn1;
        if (!(GAOT_FFN_ABL & 8) || ks == 15) {
            const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
            hn0 = *reinterpret_cast<const bf16x8*>(hb + slot);
            hn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + slot);
        }
        const bf16x8 wa = __builtin_bit_cast(bf16x8, w[ks & 3]), wg = __builtin_bit_cast(bf16x8, w[4 + (ks & 3)]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h0, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h1, acc[1][1], 0, 0, 0);
    };
    hn0 = *reinterpret_cast<const bf16x8*>(hb + ((hf ^ sw) << 4));
    hn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + ((hf ^ sw) << 4));
    // ---- prologue: a | g of chunk 0 ----
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st + LA < 4) wload(wr[(st + LA) % RD], 0, st + LA);
        else prefetch_body(0, st + LA - 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) ag_kstep(agc, wr[st % RD], 4 * st + s);
    }

    FFN_STAMP(2);
    for (int c = 0; c < NC; ++c) {
        char* ub = lds + H_BYTES + (c & 1) * U_BYTES;
        char* agb = lds + H_BYTES + 2 * U_BYTES + (c & 1) * AG_BYTES;
        if (c == NC - 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + 32 * i + l31;
                const unsigned roff = (m < M && rsrc_p) ? (unsigned)m * (unsigned)ldres * 4u : 0x80000000u;     // out of range -> zeros
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rres[i][jt][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0));
            }
        }
        // ---- a | g of chunk c + 1 (MFMAs) with the epilogue of chunk c (vector instructions) between its k-steps ----
        // epilogue of chunk c: a | g rounded to bf16 (-> HBM), u = silu(a) g from the ROUNDED values (-> HBM, -> LDS).
        // agc[jt][i][r]: column c*128 + 32 wave + mfma32_row(r, hf) of a (jt 0) / g (jt 1), row m0 + 32 i + l31.  Work unit (i, q) = the
        // lane's four columns 8q + 4hf .. + 3 of row tile i; the units (i, q) and (i, q + 2) leave together (half-wave pairing: 16 bytes
        // per lane).  16 k-steps: even k-step 2j runs unit j (order: (0,0) (0,2) (0,1) (0,3) (1,0) ...), odd k-steps 4j' + 3 pair and store.
        unsigned pa[2][2][2], pg[2][2][2], pu[2][2][2];      // [slot of the pair][unit within the pair][2 words]
        const int ncol = c * FC + wave * 32;
        auto half_unit = [&](int i, int q, int ps, int pw, int h) {       // elements 2h, 2h + 1 of unit (i, q)
            float av[2], gv[2];
            pa[ps][pw][h] = pack2(agc[0][i][4 * q + 2 * h], agc[0][i][4 * q + 2 * h + 1]);
            pg[ps][pw][h] = pack2(agc[1][i][4 * q + 2 * h], agc[1][i][4 * q + 2 * h + 1]);
            // the rounded values back as fp32: low / high half of the packed word
            av[0] = __uint_as_float(pa[ps][pw][h] << 16); av[1] = __uint_as_float(pa[ps][pw][h] & 0xffff0000u);
            gv[0] = __uint_as_float(pg[ps][pw][h] << 16); gv[1] = __uint_as_float(pg[ps][pw][h] & 0xffff0000u);
            // breadth first: no instruction reads its predecessor's result (a dependent v_exp_f32 / v_rcp_f32 chain stalls the wave's
            // one issue port for the latency of each link, and with it the MFMAs queued behind)
            float o[2], t[2];
            if (GAOT_FFN_ABL & 4) {
                o[0] = av[0]; o[1] = av[1];
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) t[e] = -1.4426950408889634f * av[e];
#pragma unroll
                for (int e = 0; e < 2; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
#pragma unroll
                for (int e = 0; e < 2; ++e) t[e] = 1.0f + t[e];
#pragma unroll
                for (int e = 0; e < 2; ++e) t[e] = __builtin_amdgcn_rcpf(t[e]);
#pragma unroll
                for (int e = 0; e < 2; ++e) o[e] = av[e] * t[e];
#pragma unroll
                for (int e = 0; e < 2; ++e) o[e] = o[e] * gv[e];
            }
            pu[ps][pw][h] = pack2(o[0], o[1]);
        };
        auto finish = [&](int i, int q, int ps) {       // units (i, q) [pw 0] and (i, q + 2) [pw 1]
            const int ml = 32 * i + l31, m = m0 + ml;
            if constexpr (SAVE) {
                // a | g leave through LDS (row ml: 512 B = 16 chunks of a, 16 of g; chunk stored at chunk ^ (row & 31)) and are written to
                // HBM as whole 256-byte row segments during the y phase: stored from this layout (16 bytes per lane, two lanes per row) a
                // wave-instruction touches 32 different 128-byte lines with 32 bytes each, and the 24 such stores per wave and chunk cost
                // the CU's address path as much as the weight stream itself (merged phase 4 460 -> 6 500 cycles)
                {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pa[ps][0][0], pa[ps][1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pa[ps][0][1], pa[ps][1][1], false, false);
                    const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                    *reinterpret_cast<u32x4*>(agb + ml * 512 + (((4 * wave + q + 2 * hf) ^ (ml & 31)) << 4)) = v;
                }
                {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pg[ps][0][0], pg[ps][1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pg[ps][0][1], pg[ps][1][1], false, false);
                    const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                    *reinterpret_cast<u32x4*>(agb + ml * 512 + (((16 + 4 * wave + q + 2 * hf) ^ (ml & 31)) << 4)) = v;
                }
            }
            const auto r0 = __builtin_amdgcn_permlane32_swap(pu[ps][0][0], pu[ps][1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pu[ps][0][1], pu[ps][1][1], false, false);
            const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            // LDS: row ml (256 B), 16-byte chunk 4 wave + q + 2 hf, stored at chunk ^ (row & 15)
            *reinterpret_cast<u32x4*>(ub + ml * 256 + (((4 * wave + q + 2 * hf) ^ (ml & 15)) << 4)) = v;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) agn[jt][i][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            prefetch_body(c, st + LA);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ks = 4 * st + s;
                __builtin_amdgcn_sched_barrier(0);      // a k-step's MFMAs, fragment reads and its share of the epilogue stay together
                ag_kstep(agn, wr[(4 + st) % RD], ks);
                {
                    const int j = ks >> 1, i = j >> 2, jj = j & 3;       // jj: 0 -> q 0, 1 -> q 2, 2 -> q 1, 3 -> q 3
                    half_unit(i, (jj >> 1) + 2 * (jj & 1), jj >> 1, jj & 1, ks & 1);
                    if ((ks & 3) == 3) finish(i, jj >> 1, jj >> 1);
                }
                // the region's order: fragment reads first, then every MFMA followed by its share of the vector work
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        FFN_STAMP(3 + 3 * c);
        if (!(GAOT_FFN_ABL & 16)) __builtin_amdgcn_s_barrier();
        FFN_STAMP(4 + 3 * c);     // the chunk's u is complete; the other buffer's readers (chunk c - 1) are all past their reads
        // ---- y += u chunk W2c^T: two 64-deep steps ----
        bf16x8 un0 = *reinterpret_cast<const bf16x8*>(ub + l31 * 256 + ((hf ^ sw) << 4)), un1 = *reinterpret_cast<const bf16x8*>(ub + (32 + l31) * 256 + ((hf ^ sw) << 4));
#pragma unroll
        for (int st = 4; st < 6; ++st) {
            prefetch_body(c, st + LA);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 u0 = un0, u1 = un1;
                if (4 * (st - 4) + s + 1 < 8 && !(GAOT_FFN_ABL & 8)) {
                    const int slot = ((2 * (4 * (st - 4) + s + 1) + hf) ^ sw) << 4;
                    un0 = *reinterpret_cast<const bf16x8*>(ub + l31 * 256 + slot);
                    un1 = *reinterpret_cast<const bf16x8*>(ub + (32 + l31) * 256 + slot);
                }
                if constexpr (SAVE) {
                    // this wave's share of the chunk's a | g (rows 16 wave .. + 15, two rows per instruction) and u (four rows per
                    // instruction) from LDS to HBM: one piece of a | g per k-step, one of u every other k-step
                    const int kk = 4 * (st - 4) + s;
                    {
                        const int row = 16 * wave + 2 * kk + hf, ch = l31, m = m0 + row;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(agb + row * 512 + ((ch ^ (row & 31)) << 4));
                        const unsigned off = m < M ? (unsigned)m * (unsigned)F * 4u + (unsigned)(((ch < 16 ? 0 : F) + c * FC + (ch & 15) * 8) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, agrs, off, 0, 0);
                    }
                    if ((kk & 1) == 0) {
                        const int row = 16 * wave + 2 * kk + (lane >> 4), ch = lane & 15, m = m0 + row;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(ub + row * 256 + ((ch ^ (row & 15)) << 4));
                        const unsigned off = m < M ? (unsigned)m * (unsigned)F * 2u + (unsigned)((c * FC + ch * 8) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, urs, off, 0, 0);
                    }
                }
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wr[(4 + st) % RD][s]), w1 = __builtin_bit_cast(bf16x8, wr[(4 + st) % RD][4 + s]);
                y[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, u0, y[0][0], 0, 0, 0);
                y[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, u0, y[1][0], 0, 0, 0);
                y[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, u1, y[0][1], 0, 0, 0);
                y[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, u1, y[1][1], 0, 0, 0);
                // the NEXT k-step's fragment reads stay in front of this k-step's MFMAs (left alone, the scheduler sinks them behind
                // the MFMAs to save two registers and every k-step then starts with an exposed LDS round trip: 2 016 -> 1 1xx cycles)
                __builtin_amdgcn_sched_group_barrier(0x100, SAVE ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) agc[jt][i] = agn[jt][i];
        FFN_STAMP(5 + 3 * c);
    }
    // ---- y (+ residual): column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31 ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31;
        const unsigned rowoff = m < M ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {y[jt][i][4 * q], y[jt][i][4 * q + 1], y[jt][i][4 * q + 2], y[jt][i][4 * q + 3]};
                if constexpr (NORM) {       // the residual is the normalised row: (h r) w, as k_rmsnorm_fwd forms it
                    const float r = rstd_l[32 * i + l31];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(NW + wave * 64 + 32 * jt + 8 * q + 4 * hf);
                    const f32x4 hv = rres[i][jt][q];
                    v += f32x4{hv[0] * r * g[0], hv[1] * r * g[1], hv[2] * r * g[2], hv[3] * r * g[3]};
                } else {
                    v += rres[i][jt][q];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, rowoff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0);
            }
    }
    FFN_STAMP(40);
}

constexpr int FWD_LDS = H_BYTES + 2 * U_BYTES, FWD_LDS_SAVE = FWD_LDS + 2 * AG_BYTES;
#ifndef GAOT_FFN_FWD_RING
#define GAOT_FFN_FWD_RING 3
#endif
constexpr int FWD_RING = GAOT_FFN_FWD_RING;

template <bool SAVE, bool NORM, bool OPROJ = false>
int launch_ffn_fwd(const void* x, const void* w13p, const void* w2p, const float* r, float* y, void* ag, void* u, int M, int F, int ldr,
                   const TailArgs& ta, hipStream_t st) {
    auto kern = k_ffn_fwd<SAVE, FWD_RING, NORM, OPROJ>;
    constexpr int LDS = (SAVE ? FWD_LDS_SAVE : FWD_LDS) + (NORM ? 256 : 0) + (OPROJ ? 1024 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            gaot_set_error("ffn_fwd: cannot set dynamic LDS %d: %s", LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(kern, dim3((unsigned)(8 * per)), dim3(256), LDS, st, (const bf16_t*)x, (const u32x4*)w13p, (const u32x4*)w2p, r, y,
                 (bf16_t*)ag, (bf16_t*)u, M, F, ldr, ta);
    return GAOT_OK;
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------------
// The backward's first half as one launch: a | g RECOMPUTED from the saved normalised input (the forward then writes neither a | g nor
// u: 96 MB per layer at S = 16 384 that never cross HBM in the forward and are not read back here), du = dy W2, the SwiGLU derivative on
// the rounded a, g, du (the arithmetic of k_swiglu_bwd_bf16), and
//     dag = d(a) | d(g)  (bf16 [M][2F]: the operand of the dx and dW13 GEMMs),   u = silu(a) g  (bf16 [M][F]: the operand of dW2),
//     dyb = bf16(dy)     ([M][256]: the operand of dW2)
// written once.  X [M][256] bf16, DY [M][256] fp32, W13p as in the forward, W2tp = fragments of W2^T (layout table above).
// Per chunk of 128 columns a wave issues 64 (a | g) + 32 (du) MFMAs; the derivative of chunk c sits between the MFMAs of chunk c + 1.
// No barrier in the chunk loop: a wave stages and stores its own 32 columns.
constexpr int STG_ROW = 208;              // a staged row: 32 da | 32 dg | 32 u bf16 = 192 B, + 16 B so that 16 consecutive rows start in 16 different bank groups
constexpr int BWD_STG = 64 * STG_ROW;     // a wave's staging
constexpr int BWD_LDS = 2 * H_BYTES + 4 * BWD_STG;

template <int RD>
__global__ __launch_bounds__(256, 1) void k_ffn_bwd(const bf16_t* __restrict__ X, const float* __restrict__ DY, const u32x4* __restrict__ W13p,
                                                     const u32x4* __restrict__ W2tp, bf16_t* __restrict__ DAG, bf16_t* __restrict__ U,
                                                     bf16_t* __restrict__ DYB, int M, int F) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB, NC = F / FC;
    constexpr int LA = RD - 1;
    static_assert(6 % RD == 0, "the ring is indexed by the step within a chunk");

    const int64_t xbytes = (int64_t)M * D * 2, dagbytes = (int64_t)M * F * 4, ubytes = (int64_t)M * F * 2, dybytes = (int64_t)M * D * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)DY, 0, (int)(dybytes > 0x7fffffff ? 0x7fffffff : dybytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dagrs = __builtin_amdgcn_make_buffer_rsrc((void*)DAG, 0, (int)(dagbytes > 0x7fffffff ? 0x7fffffff : dagbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)(ubytes > 0x7fffffff ? 0x7fffffff : ubytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dybrs = __builtin_amdgcn_make_buffer_rsrc((void*)DYB, 0, DYB ? (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes) : 0, 0x00020000);

    // steps of chunk c: st 0..3 = the four 64-deep k-slices of h W13c^T (8 blocks: a, g x 4 k-steps), st 4, 5 = the two 128-deep k-slices
    // of dy W2tc^T (8 blocks: 8 k-steps of the one 32-column tile); 16 MFMAs per step either way
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t w13rs = __builtin_amdgcn_make_buffer_rsrc((void*)W13p, 0, 2 * F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2trs = __builtin_amdgcn_make_buffer_rsrc((void*)W2tp, 0, D * F * 2, 0x00020000);
    auto wload = [&](u32x4 (&dst)[8], int c, int st) {
        if (st < 4) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    dst[jt * 4 + s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w13rs, lane * 16, ((((c * 4 + wv) * 2 + jt) * 16) + 4 * st + s) * 1024, 0));
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                dst[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w2trs, lane * 16, (((c * 4 + wv) * 16) + 8 * (st - 4) + s) * 1024, 0));
        }
    };
    u32x4 wr[RD][8];
    auto prefetch = [&](int c, int tpos) {     // tpos = step of chunk c, may run into chunk c + 1
        wload(wr[tpos % RD], tpos < 6 ? c : c + 1, tpos < 6 ? tpos : tpos - 6);
    };
#pragma unroll
    for (int p = 0; p < LA; ++p) prefetch(0, p);

    // tiles: h (bf16 as stored) by LDS-DMA; dy fp32 -> bf16 through registers; both 64 rows x 512 B, slot s of row r = chunk s ^ (r & 15)
    {
        const int lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = 2 * i + lh;
            const int voff = (m0 + wave * 16 + x) * D * 2 + (((lane & 31) ^ x) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + i) * 1024), 16, voff, 0, 0, 0);
        }
        f32x4 d0[8], d1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // row 8 i + (tid >> 5), 16-byte bf16 chunk tid & 31 = 8 floats
            const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
            const unsigned off = m < M ? (unsigned)m * (unsigned)D * 4u + (unsigned)(threadIdx.x & 31) * 32u : 0x80000000u;
            d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, off, 0, 0));
            d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, off, 16, 0));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31, m = m0 + row;
            const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
            *reinterpret_cast<u32x4*>(lds + H_BYTES + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
            if (DYB) __builtin_amdgcn_raw_buffer_store_b128(v, dybrs, m < M ? (unsigned)m * (unsigned)D * 2u + (unsigned)ch * 16u : 0x80000000u, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16 agc[2][2], agn[2][2], duc[2], dun[2];
    const int sw = l31 & 15;
    const char* hb = lds + l31 * 512;                       // h rows l31 (i = 0) and 32 + l31 (+ 32 * 512)
    const char* yb = lds + H_BYTES + l31 * 512;             // dy rows, same layout
    char* stg = lds + 2 * H_BYTES + wave * BWD_STG;         // this wave's staging: row r at r * STG_ROW: da 64 B | dg 64 B | u 64 B
    bf16x8 fn0, fn1;
    // k-step ks (0..15) of the a | g product (fragments of h), or of the du product (fragments of dy), with the NEXT k-step's two
    // activation fragments requested first; `nxt` = base of the tile the next k-step reads (h or dy)
    auto ag_kstep = [&](f32x16 (&acc)[2][2], const u32x4 (&w)[8], int ks, const char* nxt) {
        const bf16x8 h0 = fn0, h1 = fn1;
        const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
        fn0 = *reinterpret_cast<const bf16x8*>(nxt + slot);
        fn1 = *reinterpret_cast<const bf16x8*>(nxt + 32 * 512 + slot);
        const bf16x8 wa = __builtin_bit_cast(bf16x8, w[ks & 3]), wg = __builtin_bit_cast(bf16x8, w[4 + (ks & 3)]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h0, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h1, acc[1][1], 0, 0, 0);
    };
    auto du_kstep = [&](f32x16 (&acc)[2], const u32x4 (&w)[8], int ks, const char* nxt) {
        const bf16x8 y0 = fn0, y1 = fn1;
        const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
        fn0 = *reinterpret_cast<const bf16x8*>(nxt + slot);
        fn1 = *reinterpret_cast<const bf16x8*>(nxt + 32 * 512 + slot);
        const bf16x8 wt = __builtin_bit_cast(bf16x8, w[ks & 7]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt, y0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt, y1, acc[1], 0, 0, 0);
    };
    auto zero = [&](f32x16 (&a)[2][2], f32x16 (&d)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[0][i][r] = 0.f; a[1][i][r] = 0.f; d[i][r] = 0.f; }
    };
    fn0 = *reinterpret_cast<const bf16x8*>(hb + ((hf ^ sw) << 4));
    fn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + ((hf ^ sw) << 4));
    // ---- prologue: a | g and du of chunk 0 ----
    zero(agc, duc);
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        prefetch(0, st + LA);
        if (st < 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ag_kstep(agc, wr[st % RD], 4 * st + s, (4 * st + s == 15) ? yb : hb);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) du_kstep(duc, wr[st % RD], 8 * (st - 4) + s, (8 * (st - 4) + s == 15) ? hb : yb);
        }
    }

    for (int c = 0; c < NC; ++c) {
        // the derivative of chunk c: agc[jt][i][r] / duc[i][r]: column c*128 + 32 wave + mfma32_row(r, hf), row m0 + 32 i + l31.  Unit (i, q)
        // = the lane's four columns 8q + 4hf .. + 3 of row tile i; the units (i, q) and (i, q + 2) leave together (16 bytes per lane).
        unsigned pa[2][2][2], pg[2][2][2], pu[2][2][2];      // [slot of the pair][unit within the pair][2 words]
        auto half_unit = [&](int i, int q, int ps, int pw, int h) {       // elements 2h, 2h + 1 of unit (i, q)
            const unsigned wa = pack2(agc[0][i][4 * q + 2 * h], agc[0][i][4 * q + 2 * h + 1]);
            const unsigned wg = pack2(agc[1][i][4 * q + 2 * h], agc[1][i][4 * q + 2 * h + 1]);
            const unsigned wd = pack2(duc[i][4 * q + 2 * h], duc[i][4 * q + 2 * h + 1]);
            float av[2], gv[2], dv[2], sg[2], t[2], da[2], dg[2], o[2];
            av[0] = __uint_as_float(wa << 16); av[1] = __uint_as_float(wa & 0xffff0000u);
            gv[0] = __uint_as_float(wg << 16); gv[1] = __uint_as_float(wg & 0xffff0000u);
            dv[0] = __uint_as_float(wd << 16); dv[1] = __uint_as_float(wd & 0xffff0000u);
            // breadth first (see the forward); the arithmetic of k_swiglu_bwd_bf16 / k_swiglu_fwd_bf16 on the rounded a, g, du
#pragma unroll
            for (int e = 0; e < 2; ++e) t[e] = -1.4426950408889634f * av[e];
#pragma unroll
            for (int e = 0; e < 2; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
#pragma unroll
            for (int e = 0; e < 2; ++e) t[e] = 1.0f + t[e];
#pragma unroll
            for (int e = 0; e < 2; ++e) sg[e] = __builtin_amdgcn_rcpf(t[e]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                da[e] = dv[e] * gv[e] * sg[e] * (1.f + av[e] * (1.f - sg[e]));
                dg[e] = dv[e] * av[e] * sg[e];
                o[e] = av[e] * sg[e] * gv[e];
            }
            pa[ps][pw][h] = pack2(da[0], da[1]);
            pg[ps][pw][h] = pack2(dg[0], dg[1]);
            pu[ps][pw][h] = pack2(o[0], o[1]);
        };
        auto finish = [&](int i, int q, int ps) {       // units (i, q) [pw 0] and (i, q + 2) [pw 1] -> 16 bytes per lane: columns 8q + 16hf .. + 7
            const int ml = 32 * i + l31, chunk = q + 2 * hf;      // 16-byte chunk 0..3 of the wave's 64-byte row segment
            char* row = stg + ml * STG_ROW + (chunk << 4);
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pa[ps][0][0], pa[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pa[ps][0][1], pa[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(row) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pg[ps][0][0], pg[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pg[ps][0][1], pg[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(row + 64) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pu[ps][0][0], pu[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pu[ps][0][1], pu[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(row + 128) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
        };
        // the wave's staged chunk to HBM: 16 rows x 64 B per instruction and matrix (lane: row lane >> 2, chunk lane & 3)
        auto copy_out = [&](int piece) {       // piece 0..11: rows 16 (piece & 3) .., matrix piece >> 2 (da, dg, u)
            const int rowl = 16 * (piece & 3) + (lane >> 2), ch = lane & 3, mat = piece >> 2, m = m0 + rowl;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + rowl * STG_ROW + mat * 64 + (ch << 4));
            const int col = c * FC + wave * 32 + ch * 8;
            if (mat < 2) __builtin_amdgcn_raw_buffer_store_b128(v, dagrs, m < M ? (unsigned)m * (unsigned)F * 4u + (unsigned)((mat * F + col) * 2) : 0x80000000u, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, urs, m < M ? (unsigned)m * (unsigned)F * 2u + (unsigned)(col * 2) : 0x80000000u, 0, 0);
        };
        zero(agn, dun);
        // 24 groups of four MFMAs (16 a | g k-steps, 8 pairs of du k-steps) of chunk c + 1; group g carries half-unit g (g < 16), the
        // pairings at g = 3, 7, 11, 15, the copy-out in groups 16..23 -- after the wave's own LDS writes, no barrier
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            prefetch(c + 1, st + LA);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const int g = 4 * st + s;
                if (st < 4) {
                    ag_kstep(agn, wr[st % RD], g, g == 15 ? yb : hb);
                    const int j = g >> 1, i = j >> 2, jj = j & 3;       // jj: 0 -> q 0, 1 -> q 2, 2 -> q 1, 3 -> q 3
                    half_unit(i, (jj >> 1) + 2 * (jj & 1), jj >> 1, jj & 1, g & 1);
                    if ((g & 3) == 3) finish(i, jj >> 1, jj >> 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                    }
                } else {
                    const int ks = 2 * (g - 16);
                    du_kstep(dun, wr[st % RD], ks, yb);
                    du_kstep(dun, wr[st % RD], ks + 1, ks + 1 == 15 ? hb : yb);
                    if (g - 16 < 6) { copy_out(2 * (g - 16)); copy_out(2 * (g - 16) + 1); }
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            agc[0][i] = agn[0][i]; agc[1][i] = agn[1][i]; duc[i] = dun[i];
        }
    }
}

// ---- RMSNorm backward as the epilogue of a product that leaves d(norm(x)) in the accumulators (reference attn.py:167-178 autograd) ----
// dn[jt][i][r] = gradient w.r.t. the normalised rows: column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31.  Writes
//   dx = r w dn - x r^3 mean(x w dn) (+ dres) (+ dtap)   (fp32 [M][256]; dres / dtap: gradients reaching x through its other consumers)
//   dwp[blk][256] = sum over the block's 64 rows of dn x r   (one partial row per workgroup, summed later in fixed order)
// A row's dot product is the sum of the four waves' 64-column partials (through LDS, wave order); the column sums walk the block's
// rows in order through a [64][260] fp32 LDS image (pitch 1 040 B: the 16-byte stores of eight consecutive rows cover all banks).
// `lds` must be free for 64 * 1040 + 1024 bytes; every wave of the workgroup calls this (two barriers inside).
constexpr int NB_PITCH = 260, NB_LDS = 64 * NB_PITCH * 4 + 1024;
// `dxtile` (may be null): the dx rows also leave as a bf16 [64][256] LDS tile (slot s of row r = chunk s ^ (r & 15)) for a product that follows.
__device__ __forceinline__ void norm_bwd_epilogue(f32x16 (&dn)[2][2], char* lds, const float* __restrict__ X, int ldx, const float* __restrict__ NW,
                                                  const float* __restrict__ RSTD, const float* __restrict__ DRES, const float* __restrict__ DTAP,
                                                  float* __restrict__ DX, float* __restrict__ DWP, int M, int m0, int blk, char* dxtile = nullptr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    float* T = reinterpret_cast<float*>(lds);
    float* part = reinterpret_cast<float*>(lds + 64 * NB_PITCH * 4);      // [4 waves][64 rows]
    f32x4 xv[2][2][4];
    float rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31, mm = m < M ? m : M - 1;
        rs[i] = RSTD[mm];
        float dot = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wave * 64 + 32 * jt + 8 * q + 4 * hf;
                const f32x4 v = *reinterpret_cast<const f32x4*>(X + (int64_t)mm * ldx + col), g = *reinterpret_cast<const f32x4*>(NW + col);
                xv[i][jt][q] = v;
                const f32x4 d = {dn[jt][i][4 * q], dn[jt][i][4 * q + 1], dn[jt][i][4 * q + 2], dn[jt][i][4 * q + 3]};
                dot += v[0] * d[0] * g[0] + v[1] * d[1] * g[1] + v[2] * d[2] * g[2] + v[3] * d[3] * g[3];
                const f32x4 tt = {d[0] * v[0] * rs[i], d[1] * v[1] * rs[i], d[2] * v[2] * rs[i], d[3] * v[3] * rs[i]};
                *reinterpret_cast<f32x4*>(T + (32 * i + l31) * NB_PITCH + col) = tt;
            }
        dot += __shfl_xor(dot, 32, 64);
        if (hf == 0) part[wave * 64 + 32 * i + l31] = dot;
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = 32 * i + l31, m = m0 + ml;
        const float dot = ((part[ml] + part[64 + ml]) + part[128 + ml]) + part[192 + ml];
        const float r = rs[i], c = dot * r * r * r / (float)D;
        if (m < M || dxtile) {      // with a tile, rows past M go through as well (finite or not, a row of the product that follows depends on its own row only)
            const bool live = m < M;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 64 + 32 * jt + 8 * q + 4 * hf;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(NW + col), v = xv[i][jt][q];
                    f32x4 o = {r * g[0] * dn[jt][i][4 * q] - v[0] * c, r * g[1] * dn[jt][i][4 * q + 1] - v[1] * c,
                               r * g[2] * dn[jt][i][4 * q + 2] - v[2] * c, r * g[3] * dn[jt][i][4 * q + 3] - v[3] * c};
                    if (DRES && live) o += *reinterpret_cast<const f32x4*>(DRES + (int64_t)m * D + col);
                    if (DTAP && live) o += *reinterpret_cast<const f32x4*>(DTAP + (int64_t)m * D + col);
                    if (live) *reinterpret_cast<f32x4*>(DX + (int64_t)m * D + col) = o;
                    if (dxtile) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2*>(dxtile + ml * 512 + (((8 * wave + 4 * jt + q) ^ (ml & 15)) << 4) + 8 * hf) = u32x2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                    }
                }
        }
    }
    {   // column sums of dn x r over the block's rows, row order (rows past M hold zeros: their dn is a product of zero-filled loads)
        float sum = 0.f;
#pragma unroll 8
        for (int row = 0; row < 64; ++row) sum += T[row * NB_PITCH + threadIdx.x];
        DWP[(int64_t)blk * D + threadIdx.x] = sum;
    }
}

// The same pass WITH the input gradient: dx = dag W13 (+ dy when the block's residual is the FFN's own input, attn.py:229) accumulated
// chunk by chunk from the dag chunk in LDS -- the dag tensor is written for the dW13 product but not read back for dx, and the
// stand-alone dx GEMM (K = 2F) is gone.  Per chunk a wave issues 64 (a | g) + 32 (du) + 64 (dx) MFMAs; ten steps, ring of five.
// The dag chunk is shared by the four waves (row = 512 B: 16 chunks da | 16 chunks dg, chunk stored at chunk ^ (row & 15)), double
// buffered, one barrier per chunk; u stays in the wave's private staging.
constexpr int UST_ROW = 80;                // a wave's staged u row: 64 B + 16 B (16 rows -> 16 bank groups)
constexpr int BWDX_LDS = 2 * H_BYTES + 2 * AG_BYTES + 4 * 64 * UST_ROW;

// NB: ffn_norm's backward in the epilogue (norm_bwd_epilogue): DX receives dh = d(un-normalised rows), DWP the norm weight's partials
struct NormBwdArgs { const float* H; int ldh; const float* NW; const float* RSTD; float* DWP; };
template <bool NB>
__global__ __launch_bounds__(256, 1) void k_ffn_bwd_dx(const bf16_t* __restrict__ X, const float* __restrict__ DY, const u32x4* __restrict__ W13p,
                                                        const u32x4* __restrict__ W2tp, const u32x4* __restrict__ W13tp, bf16_t* __restrict__ DAG,
                                                        bf16_t* __restrict__ U, bf16_t* __restrict__ DYB, float* __restrict__ DX, int M, int F,
                                                        int add_dy, NormBwdArgs nb) {
    constexpr int RD = 5, LA = 3;      // five static slots, four live at a time (three steps in flight + the one in use)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB, NC = F / FC;

    const int64_t xbytes = (int64_t)M * D * 2, dagbytes = (int64_t)M * F * 4, ubytes = (int64_t)M * F * 2, dybytes = (int64_t)M * D * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)DY, 0, (int)(dybytes > 0x7fffffff ? 0x7fffffff : dybytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dagrs = __builtin_amdgcn_make_buffer_rsrc((void*)DAG, 0, (int)(dagbytes > 0x7fffffff ? 0x7fffffff : dagbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)(ubytes > 0x7fffffff ? 0x7fffffff : ubytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dybrs = __builtin_amdgcn_make_buffer_rsrc((void*)DYB, 0, DYB ? (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dxrs = __builtin_amdgcn_make_buffer_rsrc((void*)DX, 0, (int)(dybytes > 0x7fffffff ? 0x7fffffff : dybytes), 0x00020000);

    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t w13rs = __builtin_amdgcn_make_buffer_rsrc((void*)W13p, 0, 2 * F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2trs = __builtin_amdgcn_make_buffer_rsrc((void*)W2tp, 0, D * F * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w13trs = __builtin_amdgcn_make_buffer_rsrc((void*)W13tp, 0, 2 * F * D * 2, 0x00020000);
    // kinds of step (8 fragment blocks, 16 MFMAs each): A(st 0..3) = k-slice st of h W13c^T; B(st 0, 1) = k-slice of dy W2tc^T;
    // C(st 0..3) = k-slice of dag_c W13c (fragments of W13^T: output columns 64 w + 32 jt .., k over the chunk's 256 dag columns)
    auto wloadA = [&](u32x4 (&dst)[8], int c, int st) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                dst[jt * 4 + s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w13rs, lane * 16, ((((c * 4 + wv) * 2 + jt) * 16) + 4 * st + s) * 1024, 0));
    };
    auto wloadB = [&](u32x4 (&dst)[8], int c, int st) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            dst[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w2trs, lane * 16, (((c * 4 + wv) * 16) + 8 * st + s) * 1024, 0));
    };
    auto wloadC = [&](u32x4 (&dst)[8], int c, int st) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                dst[jt * 4 + s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w13trs, lane * 16, ((((c * 4 + wv) * 2 + jt) * 16) + 4 * st + s) * 1024, 0));
    };
    // the wave's timeline: positions 0..5 = steps A0..A3, B0, B1 of chunk 0 (prologue); iteration c of the loop = ten positions
    // b = 0..3: A0..A3 of chunk c + 1 (beside the derivative of chunk c), b = 4..7: C0..C3 of chunk c, b = 8, 9: B0, B1 of chunk c + 1
    // (into the du accumulators the derivative has just finished with: no second set).  Ring slot = position % 5 (10 % 5 == 0: static).
    u32x4 wr[RD][8];
    auto load_chunk_step = [&](u32x4 (&dst)[8], int c, int st) {     // st 0..5 of the A / B sequence of chunk c
        if (st < 4) wloadA(dst, c, st);
        else wloadB(dst, c, st - 4);
    };
    auto prefetch_body = [&](int c, int tpos) {     // tpos = body position of iteration c, may run into iteration c + 1
        const int cc = tpos < 10 ? c : c + 1, b = tpos < 10 ? tpos : tpos - 10;
        if (b < 4) wloadA(wr[(6 + tpos) % RD], cc + 1, b);
        else if (b < 8) wloadC(wr[(6 + tpos) % RD], cc, b - 4);
        else wloadB(wr[(6 + tpos) % RD], cc + 1, b - 8);
    };
#pragma unroll
    for (int p = 0; p < LA; ++p) load_chunk_step(wr[p % RD], 0, p);

    // tiles: h (bf16 as stored) by LDS-DMA; dy fp32 -> bf16 through registers; both 64 rows x 512 B, slot s of row r = chunk s ^ (r & 15)
    {
        const int lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = 2 * i + lh;
            const int voff = (m0 + wave * 16 + x) * D * 2 + (((lane & 31) ^ x) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + i) * 1024), 16, voff, 0, 0, 0);
        }
        f32x4 d0[8], d1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
            const unsigned off = m < M ? (unsigned)m * (unsigned)D * 4u + (unsigned)(threadIdx.x & 31) * 32u : 0x80000000u;
            d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, off, 0, 0));
            d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, off, 16, 0));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31, m = m0 + row;
            const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
            *reinterpret_cast<u32x4*>(lds + H_BYTES + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
            if (DYB) __builtin_amdgcn_raw_buffer_store_b128(v, dybrs, m < M ? (unsigned)m * (unsigned)D * 2u + (unsigned)ch * 16u : 0x80000000u, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16 agc[2][2], agn[2][2], duc[2], dx[2][2];
    const int sw = l31 & 15;
    const char* hb = lds + l31 * 512;
    const char* yb = lds + H_BYTES + l31 * 512;
    char* ust = lds + 2 * H_BYTES + 2 * AG_BYTES + wave * 64 * UST_ROW;      // this wave's staged u: row r at r * UST_ROW
    bf16x8 fn0, fn1;
    auto ag_kstep = [&](f32x16 (&acc)[2][2], const u32x4 (&w)[8], int ks, const char* nxt) {
        const bf16x8 h0 = fn0, h1 = fn1;
        const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
        fn0 = *reinterpret_cast<const bf16x8*>(nxt + slot);
        fn1 = *reinterpret_cast<const bf16x8*>(nxt + 32 * 512 + slot);
        const bf16x8 wa = __builtin_bit_cast(bf16x8, w[ks & 3]), wg = __builtin_bit_cast(bf16x8, w[4 + (ks & 3)]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h0, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, h1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg, h1, acc[1][1], 0, 0, 0);
    };
    auto du_kstep = [&](f32x16 (&acc)[2], const u32x4 (&w)[8], int ks, const char* nxt) {
        const bf16x8 y0 = fn0, y1 = fn1;
        const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
        fn0 = *reinterpret_cast<const bf16x8*>(nxt + slot);
        fn1 = *reinterpret_cast<const bf16x8*>(nxt + 32 * 512 + slot);
        const bf16x8 wt = __builtin_bit_cast(bf16x8, w[ks & 7]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt, y0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt, y1, acc[1], 0, 0, 0);
    };
    auto zero = [&](f32x16 (&a)[2][2], f32x16 (&d)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[0][i][r] = 0.f; a[1][i][r] = 0.f; d[i][r] = 0.f; }
    };
    // dx starts as dy (the gradient through the block's residual) or as zero: the accumulators themselves hold the fp32 dy rows of this
    // lane's outputs -- column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31 -- no second set of registers for the residual
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31;
        const unsigned roff = (m < M && add_dy) ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;     // out of range -> zeros
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyrs, roff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0));
                dx[jt][i][4 * q] = v[0]; dx[jt][i][4 * q + 1] = v[1]; dx[jt][i][4 * q + 2] = v[2]; dx[jt][i][4 * q + 3] = v[3];
            }
    }
    fn0 = *reinterpret_cast<const bf16x8*>(hb + ((hf ^ sw) << 4));
    fn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + ((hf ^ sw) << 4));
    // ---- prologue: a | g and du of chunk 0 ----
    zero(agc, duc);
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        if (st + LA < 6) load_chunk_step(wr[(st + LA) % RD], 0, st + LA);
        else prefetch_body(0, st + LA - 6);
        if (st < 4) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ag_kstep(agc, wr[st % RD], 4 * st + s, (4 * st + s == 15) ? yb : hb);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) du_kstep(duc, wr[st % RD], 8 * (st - 4) + s, (8 * (st - 4) + s == 15) ? hb : yb);
        }
    }

    for (int c = 0; c < NC; ++c) {
        char* dagb = lds + 2 * H_BYTES + (c & 1) * AG_BYTES;
        unsigned pa[2][2][2], pg[2][2][2], pu[2][2][2];
        auto half_unit = [&](int i, int q, int ps, int pw, int h) {
            const unsigned wa = pack2(agc[0][i][4 * q + 2 * h], agc[0][i][4 * q + 2 * h + 1]);
            const unsigned wg = pack2(agc[1][i][4 * q + 2 * h], agc[1][i][4 * q + 2 * h + 1]);
            const unsigned wd = pack2(duc[i][4 * q + 2 * h], duc[i][4 * q + 2 * h + 1]);
            float av[2], gv[2], dv[2], sg[2], t2[2], da[2], dg[2], o[2];
            av[0] = __uint_as_float(wa << 16); av[1] = __uint_as_float(wa & 0xffff0000u);
            gv[0] = __uint_as_float(wg << 16); gv[1] = __uint_as_float(wg & 0xffff0000u);
            dv[0] = __uint_as_float(wd << 16); dv[1] = __uint_as_float(wd & 0xffff0000u);
#pragma unroll
            for (int e = 0; e < 2; ++e) t2[e] = -1.4426950408889634f * av[e];
#pragma unroll
            for (int e = 0; e < 2; ++e) t2[e] = __builtin_amdgcn_exp2f(t2[e]);
#pragma unroll
            for (int e = 0; e < 2; ++e) t2[e] = 1.0f + t2[e];
#pragma unroll
            for (int e = 0; e < 2; ++e) sg[e] = __builtin_amdgcn_rcpf(t2[e]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                da[e] = dv[e] * gv[e] * sg[e] * (1.f + av[e] * (1.f - sg[e]));
                dg[e] = dv[e] * av[e] * sg[e];
                o[e] = av[e] * sg[e] * gv[e];
            }
            pa[ps][pw][h] = pack2(da[0], da[1]);
            pg[ps][pw][h] = pack2(dg[0], dg[1]);
            pu[ps][pw][h] = pack2(o[0], o[1]);
        };
        auto finish = [&](int i, int q, int ps) {
            const int ml = 32 * i + l31;
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pa[ps][0][0], pa[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pa[ps][0][1], pa[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(dagb + ml * 512 + (((4 * wave + q + 2 * hf) ^ (ml & 15)) << 4)) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pg[ps][0][0], pg[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pg[ps][0][1], pg[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(dagb + ml * 512 + (((16 + 4 * wave + q + 2 * hf) ^ (ml & 15)) << 4)) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
            {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pu[ps][0][0], pu[ps][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pu[ps][0][1], pu[ps][1][1], false, false);
                *reinterpret_cast<u32x4*>(ust + ml * UST_ROW + ((q + 2 * hf) << 4)) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { agn[0][i][r] = 0.f; agn[1][i][r] = 0.f; }
        // ---- a | g of chunk c + 1 (16 k-steps of four MFMAs) with the derivative of chunk c between them ----
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            prefetch_body(c, st + LA);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const int g = 4 * st + s;
                ag_kstep(agn, wr[(6 + st) % RD], g, hb);        // (its request at k-step 15 wraps to h's first fragments: unused)
                const int j = g >> 1, i = j >> 2, jj = j & 3;
                half_unit(i, (jj >> 1) + 2 * (jj & 1), jj >> 1, jj & 1, g & 1);
                if ((g & 3) == 3) finish(i, jj >> 1, jj >> 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();        // the chunk's dag is complete; the other buffer's readers (chunk c - 1) are all past their reads
        // ---- dx += dag_c W13c: four 64-deep steps over the chunk's 256 dag columns; the chunk's dag and u rows to HBM between them ----
        const char* gb = dagb + l31 * 512;
        fn0 = *reinterpret_cast<const bf16x8*>(gb + ((hf ^ sw) << 4));
        fn1 = *reinterpret_cast<const bf16x8*>(gb + 32 * 512 + ((hf ^ sw) << 4));
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            prefetch_body(c, 4 + st + LA);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ks = 4 * st + s;
                const bf16x8 g0 = fn0, g1 = fn1;
                {       // the next k-step's fragments: dag, or (last k-step) the first dy fragments of the du product that follows
                    const char* nb = ks + 1 < 16 ? gb : yb;
                    const int slot = ((2 * ((ks + 1) & 15) + hf) ^ sw) << 4;
                    fn0 = *reinterpret_cast<const bf16x8*>(nb + slot);
                    fn1 = *reinterpret_cast<const bf16x8*>(nb + 32 * 512 + slot);
                }
                if ((ks & 1) == 0) {       // this wave's share of the chunk's dag rows (16 wave .. + 15, two rows per instruction) to HBM
                    const int kk = ks >> 1, row = 16 * wave + 2 * kk + hf, ch = l31, m = m0 + row;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(dagb + row * 512 + ((ch ^ (row & 15)) << 4));
                    const unsigned off = m < M ? (unsigned)m * (unsigned)F * 4u + (unsigned)(((ch < 16 ? 0 : F) + c * FC + (ch & 15) * 8) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, dagrs, off, 0, 0);
                } else if ((ks & 3) == 1) {      // and its own u chunk: 16 rows x 64 B per instruction
                    const int rowl = 16 * (ks >> 2) + (lane >> 2), ch = lane & 3, m = m0 + rowl;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ust + rowl * UST_ROW + (ch << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(v, urs, m < M ? (unsigned)m * (unsigned)F * 2u + (unsigned)((c * FC + wave * 32 + ch * 8) * 2) : 0x80000000u, 0, 0);
                }
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wr[(10 + st) % RD][s]), w1 = __builtin_bit_cast(bf16x8, wr[(10 + st) % RD][4 + s]);
                dx[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, g0, dx[0][0], 0, 0, 0);
                dx[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, g0, dx[1][0], 0, 0, 0);
                dx[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, g1, dx[0][1], 0, 0, 0);
                dx[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, g1, dx[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- du of chunk c + 1 (the derivative of chunk c is done with these accumulators) ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) duc[i][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            prefetch_body(c, 8 + st + LA);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                du_kstep(duc, wr[(14 + st) % RD], 8 * st + s, (8 * st + s == 15) ? hb : yb);
                if (s & 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            agc[0][i] = agn[0][i]; agc[1][i] = agn[1][i];
        }
    }
    if constexpr (NB) {
        __builtin_amdgcn_s_barrier();       // every wave is done with the tiles: the LDS is the epilogue's
        norm_bwd_epilogue(dx, lds, nb.H, nb.ldh, nb.NW, nb.RSTD, nullptr, nullptr, DX, nb.DWP, M, m0, t);
        return;
    }
    // ---- dx (+ dy): column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31 ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31;
        const unsigned rowoff = m < M ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {dx[jt][i][4 * q], dx[jt][i][4 * q + 1], dx[jt][i][4 * q + 2], dx[jt][i][4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dxrs, rowoff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0);
            }
    }
}

template <int RD>
int launch_ffn_bwd(const void* x, const float* dy, const void* w13p, const void* w2tp, void* dag, void* u, void* dyb, int M, int F, hipStream_t st) {
    auto kern = k_ffn_bwd<RD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
        if (e != hipSuccess) {
            gaot_set_error("ffn_bwd: cannot set dynamic LDS %d: %s", BWD_LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(kern, dim3((unsigned)(8 * per)), dim3(256), BWD_LDS, st, (const bf16_t*)x, dy, (const u32x4*)w13p, (const u32x4*)w2tp, (bf16_t*)dag,
                 (bf16_t*)u, (bf16_t*)dyb, M, F);
    return GAOT_OK;
}

}  // namespace

// bytes of the four packed images of one FFN's weights (forward: w13p, w2p; backward: w2tp, w13tp), each 16-byte aligned:
// offsets 0, 2F*256*2, +256*F*2, +256*F*2; total 2 * (2F*256 + 256*F) * 2 bytes
extern "C" int64_t gaot_ffn_packed_bytes(int F, int with_backward) {
    const int64_t fwd = ((int64_t)2 * F * D + (int64_t)D * F) * 2;
    return with_backward ? 2 * fwd : fwd;
}

// Pack the FFN's fp32 weights (w13 = the co-located [w1; w3], [2F][256]; w2 [256][F]) into the fragment-ordered bf16 images the fused
// kernels stream (see the layout table above).  packed: gaot_ffn_packed_bytes(F, with_backward) bytes.  One launch per FFN per step
// (the weights change with every optimizer step); stands in for the bf16 casts / transposes of the unfused path.
extern "C" int gaot_ffn_pack_multi(const gaot_ffn_pack_t* items, int num, int F, int with_backward, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(items && num > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    const int64_t frags = ((int64_t)2 * F * D + (int64_t)D * F) / 8 * (with_backward ? 2 : 1);
    for (int i0 = 0; i0 < num; i0 += PACK_MAX) {
        PackTable t{};
        const int n = std::min(PACK_MAX, num - i0);
        for (int i = 0; i < n; ++i) {
            const gaot_ffn_pack_t& it = items[i0 + i];
            GAOT_CHECK_ARG(it.w13 && it.w2 && it.packed && ((uintptr_t)it.packed % 16) == 0, "null or misaligned pointer in the table");
            t.w13[i] = it.w13; t.w2[i] = it.w2; t.wo[i] = nullptr; t.packed[i] = (bf16_t*)it.packed;
        }
        GAOT_KLAUNCH(k_ffn_pack, dim3((unsigned)std::min<int64_t>(ceil_div(frags, 256), 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t, F,
                     with_backward);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_ffn_pack(const float* w13, const float* w2, int F, void* packed, int with_backward, gaot_stream_t stream) {
    const gaot_ffn_pack_t it{w13, w2, packed};
    return gaot_ffn_pack_multi(&it, 1, F, with_backward, stream);
}

// y = w2(silu(w1 x) * w3 x) + residual in one launch: x [rows][256] bf16, packed = gaot_ffn_pack's image, residual fp32 [rows][ldr] or
// NULL, y fp32 [rows][256]; ag (bf16 [rows][2F] = w1 x | w3 x) and u (bf16 [rows][F] = silu(a) g) are written for the backward when
// both are non-NULL (both NULL: neither is written).  Values: bit-identical to gaot_ffn_w13_swiglu + gaot_gemm_ex(u, w2, residual).
extern "C" int gaot_ffn_fwd(const void* x_bf16, const void* packed, const float* residual, int64_t ldr, float* y, void* ag, void* u,
                            int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(x_bf16 && packed && y && rows > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG((ag == nullptr) == (u == nullptr), "ag and u: both or neither");
    GAOT_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)ag % 16) == 0 &&
                   ((uintptr_t)u % 16) == 0 && ((uintptr_t)residual % 16) == 0 && (!residual || ldr % 4 == 0), "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff || (residual && rows * ldr * 4 >= 0x7fffffff)) {
        gaot_set_error("gaot_ffn_fwd: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const void* w13p = p;
    const void* w2p = p + (int64_t)2 * F * D;
    const int rc = ag ? launch_ffn_fwd<true, false>(x_bf16, w13p, w2p, residual, y, ag, u, (int)rows, F, (int)ldr, TailArgs{}, (hipStream_t)stream)
                      : launch_ffn_fwd<false, false>(x_bf16, w13p, w2p, residual, y, ag, u, (int)rows, F, (int)ldr, TailArgs{}, (hipStream_t)stream);
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// The first half of the FFN's backward in one launch, for a forward that saved nothing (gaot_ffn_fwd with ag = u = NULL): x [rows][256]
// bf16 (the forward's input), dy fp32 [rows][256], packed = gaot_ffn_pack's image WITH the backward images ->
// dag = d(a) | d(g) (bf16 [rows][2F]), u = silu(a) g (bf16 [rows][F]), dyb = bf16(dy) ([rows][256], optional).  Values: those of
// gaot_ffn_w13_swiglu + gaot_gemm_ex (du, bf16) + gaot_swiglu_bwd_bf16 + gaot_cast_bf16.  Reference: autograd of attn.py:155-157.
extern "C" int gaot_ffn_bwd_dag(const void* x_bf16, const float* dy, const void* packed, void* dag, void* u, void* dyb, int64_t rows, int F,
                                gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(x_bf16 && dy && packed && dag && u && rows > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dag % 16) == 0 &&
                   ((uintptr_t)u % 16) == 0 && ((uintptr_t)dyb % 16) == 0, "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_ffn_bwd_dag: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const void* w13p = p;
    const void* w2tp = p + (int64_t)2 * F * D + (int64_t)D * F;
    const int rc = launch_ffn_bwd<3>(x_bf16, dy, w13p, w2tp, dag, u, dyb, (int)rows, F, (hipStream_t)stream);
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// The same with the input gradient computed in the launch: dx fp32 [rows][256] = dag W13 (+ dy when add_dy: the block's residual is the
// FFN's own input, reference attn.py:229).  Stands in for gaot_ffn_bwd_dag followed by gaot_gemm_ex(dag, W13^T, residual = dy).
extern "C" int gaot_ffn_bwd(const void* x_bf16, const float* dy, const void* packed, void* dag, void* u, void* dyb, float* dx, int add_dy,
                            int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(x_bf16 && dy && packed && dag && u && dx && rows > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dag % 16) == 0 &&
                   ((uintptr_t)u % 16) == 0 && ((uintptr_t)dyb % 16) == 0 && ((uintptr_t)dx % 16) == 0, "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_ffn_bwd: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_bwd_dx<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWDX_LDS);
        if (e != hipSuccess) {
            gaot_set_error("ffn_bwd: cannot set dynamic LDS %d: %s", BWDX_LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const bf16_t* w2tp = p + (int64_t)2 * F * D + (int64_t)D * F;
    const bf16_t* w13tp = w2tp + (int64_t)D * F;
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_ffn_bwd_dx<false>, dim3((unsigned)(8 * per)), dim3(256), BWDX_LDS, (hipStream_t)stream, (const bf16_t*)x_bf16, dy, (const u32x4*)p,
                 (const u32x4*)w2tp, (const u32x4*)w13tp, (bf16_t*)dag, (bf16_t*)u, (bf16_t*)dyb, dx, (int)rows, F, add_dy, NormBwdArgs{});
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// RMSNorm + FFN + residual of a Transformer block in ONE launch (reference attn.py:227-229: h = ffn_norm(h); h + ffn(h) -- the residual
// is the NORMALISED h): h fp32 [rows][ldh], norm_weight fp32 [256] -> y fp32 [rows][256] = n + w2(silu(w1 n) * w3 n) with n = RMSNorm(h);
// yb = bf16(n) ([rows][256]: the backward's input) and rstd [rows] are written for the backward (gaot_ffn_bwd, then gaot_rmsnorm_bwd on
// its dx); nothing else is saved.  Values: those of gaot_rmsnorm_fwd followed by gaot_ffn_fwd(residual = its fp32 output).
extern "C" int gaot_norm_ffn_fwd(const float* h, int64_t ldh, const float* norm_weight, float eps, const void* packed, float* y, void* yb,
                                 float* rstd, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(h && norm_weight && packed && y && yb && rstd && rows > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)yb % 16) == 0 &&
                   ((uintptr_t)norm_weight % 16) == 0 && ldh % 4 == 0 && ldh >= D, "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff || rows * ldh * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_norm_ffn_fwd: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const TailArgs ta{norm_weight, eps, (bf16_t*)yb, rstd, nullptr, 0, nullptr, nullptr};
    const int rc = launch_ffn_fwd<false, true>(nullptr, p, p + (int64_t)2 * F * D, h, y, nullptr, nullptr, (int)rows, F, (int)ldh, ta, (hipStream_t)stream);
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// ---- the whole tail of a Transformer block in one launch (reference attn.py:127, 226-229): h = x + o_proj(attn_out); n = ffn_norm(h);
// y = n + w2(silu(w1 n) * w3 n).  gaot_block_pack_multi: gaot_ffn_pack_multi WITH the backward images plus the fragment image of
// o_proj.weight ([256][256]) behind them (gaot_block_packed_bytes(F) bytes per block).
extern "C" int64_t gaot_block_packed_bytes(int F) { return gaot_ffn_packed_bytes(F, 1) + 2 * (int64_t)D * D * 2; }      // + o_proj, o_proj^T

extern "C" int gaot_block_pack_multi(const gaot_block_pack_t* items, int num, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(items && num > 0 && F > 0 && F % FC == 0, "bad argument (F must be a multiple of 128)");
    const int64_t frags = ((int64_t)2 * F * D + (int64_t)D * F) / 8 * 2;
    for (int i0 = 0; i0 < num; i0 += PACK_MAX) {
        PackTable t{};
        const int n = std::min(PACK_MAX, num - i0);
        for (int i = 0; i < n; ++i) {
            const gaot_block_pack_t& it = items[i0 + i];
            GAOT_CHECK_ARG(it.w13 && it.w2 && it.wo && it.packed && ((uintptr_t)it.packed % 16) == 0, "null or misaligned pointer in the table");
            t.w13[i] = it.w13; t.w2[i] = it.w2; t.wo[i] = it.wo; t.packed[i] = (bf16_t*)it.packed;
        }
        GAOT_KLAUNCH(k_ffn_pack, dim3((unsigned)std::min<int64_t>(ceil_div(frags, 256), 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t, F, 1);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// attn_out fp32 [rows][ldo] (the attention kernels' output, heads concatenated), x fp32 [rows][ldx] (the block's input: the first
// residual), packed = one block image of gaot_block_pack_multi -> h fp32 [rows][256] (kept for the backward's norm), y fp32 [rows][256],
// yb = bf16(ffn_norm(h)), rstd.  Values: gaot_gemm_ex(attn_out, Wo, residual = x) bit for bit, then gaot_norm_ffn_fwd to rounding.
extern "C" int gaot_block_tail_fwd(const float* attn_out, int64_t ldo, const float* x, int64_t ldx, const float* norm_weight, float eps,
                                   const void* packed, float* h, float* y, void* yb, float* rstd, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(attn_out && norm_weight && packed && h && y && yb && rstd && rows > 0 && F > 0 && F % FC == 0,
                   "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG(((uintptr_t)attn_out % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)h % 16) == 0 &&
                   ((uintptr_t)y % 16) == 0 && ((uintptr_t)yb % 16) == 0 && ((uintptr_t)norm_weight % 16) == 0 && ldo % 4 == 0 && ldo >= D &&
                   (!x || (ldx % 4 == 0 && ldx >= D)), "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff || rows * ldo * 4 >= 0x7fffffff || (x && rows * ldx * 4 >= 0x7fffffff)) {
        gaot_set_error("gaot_block_tail_fwd: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const bf16_t* wop = p + ((int64_t)2 * F * D + (int64_t)D * F) * 2;
    const TailArgs ta{norm_weight, eps, (bf16_t*)yb, rstd, attn_out, (int)ldo, (const u32x4*)wop, h};
    const int rc = launch_ffn_fwd<false, true, true>(nullptr, p, p + (int64_t)2 * F * D, x, y, nullptr, nullptr, (int)rows, F, (int)ldx, ta, (hipStream_t)stream);
    if (rc != GAOT_OK) return rc;
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// ---- the o_proj backward's input gradient written straight as the attention backward's operands (reference attn.py:127 autograd, in
// front of F.scaled_dot_product_attention's: attn.py:122-126): d_o = dh Wo never exists in fp32 -- a workgroup's 64 rows leave as the
// bf16 dO image (row-major [rows][H*32]) and as delta[b][head][s] = sum over the head's 32 columns of d_o * o (the row constants of the
// flash backward), formed from the fp32 accumulators.  Stands in for gaot_gemm_ex (d_o) + the preparation pass of gaot_attn_bwd_bf16
// (phase 1: a read of d_o and o, a write of the image).
namespace {
__global__ __launch_bounds__(256, 2) void k_oproj_bwd(const float* __restrict__ DH, const float* __restrict__ O, const u32x4* __restrict__ WoTp,
                                                       bf16_t* __restrict__ DOB, float* __restrict__ DELTA, int M, int S, int H) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB;
    const int64_t fbytes = (int64_t)M * D * 4, ibytes = (int64_t)M * D * 2;
    const __amdgpu_buffer_rsrc_t dhrs = __builtin_amdgcn_make_buffer_rsrc((void*)DH, 0, (int)(fbytes > 0x7fffffff ? 0x7fffffff : fbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)O, 0, (int)(fbytes > 0x7fffffff ? 0x7fffffff : fbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t dobrs = __builtin_amdgcn_make_buffer_rsrc((void*)DOB, 0, (int)(ibytes > 0x7fffffff ? 0x7fffffff : ibytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)WoTp, 0, D * D * 2, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    u32x4 wo[2][8];
    auto woload = [&](u32x4 (&dst)[8], int st) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, (((wv * 2 + jt) * 16) + 4 * st + s2) * 1024, 0));
    };
    woload(wo[0], 0);
    woload(wo[1], 1);
    {       // the dh rows fp32 -> bf16 -> LDS tile (row 8 i + tid >> 5, 16-byte bf16 chunk tid & 31 = 8 floats)
        f32x4 d0[8], d1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
            const unsigned off = m < M ? (unsigned)m * (unsigned)D * 4u + (unsigned)(threadIdx.x & 31) * 32u : 0x80000000u;
            d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dhrs, off, 0, 0));
            d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dhrs, off, 16, 0));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31;
            const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
            *reinterpret_cast<u32x4*>(lds + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
        }
    }
    // the attention output at this lane's positions (column 64 wave + 32 jt + 8 q + 4 hf .., row m0 + 32 i + l31): requested now
    f32x4 ov[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31;
        const unsigned roff = m < M ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ov[i][jt][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, roff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0));
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
    __builtin_amdgcn_s_barrier();
    const int sw = l31 & 15;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int slot = ((2 * (4 * st + s2) + hf) ^ sw) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(lds + l31 * 512 + slot), a1 = *reinterpret_cast<const bf16x8*>(lds + (32 + l31) * 512 + slot);
            const bf16x8 w0 = __builtin_bit_cast(bf16x8, wo[st & 1][s2]), w1 = __builtin_bit_cast(bf16x8, wo[st & 1][4 + s2]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, acc[1][1], 0, 0, 0);
        }
        if (st + 2 < 4) woload(wo[st & 1], st + 2);
    }
    // acc[jt][i][r]: head 2 wave + jt, its column mfma32_row(r, hf), row m0 + 32 i + l31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + l31;
        const unsigned rowoff = m < M ? (unsigned)m * (unsigned)D * 2u : 0x80000000u;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            float dl = 0.f;
            unsigned pk[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 o4 = ov[i][jt][q];
                dl += acc[jt][i][4 * q] * o4[0] + acc[jt][i][4 * q + 1] * o4[1] + acc[jt][i][4 * q + 2] * o4[2] + acc[jt][i][4 * q + 3] * o4[3];
                pk[q][0] = pack2(acc[jt][i][4 * q], acc[jt][i][4 * q + 1]);
                pk[q][1] = pack2(acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]);
            }
            dl += __shfl_xor(dl, 32, 64);
            if (hf == 0 && m < M) DELTA[((int64_t)(m / S) * H + 2 * wave + jt) * S + (m % S)] = dl;
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // runs (q, q+2): lower lanes end with columns 8q..8q+7, upper lanes with 16+8q..
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 2][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 2][1], false, false);
                const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(v, dobrs, rowoff + (wave * 64 + 32 * jt + 8 * q + 16 * hf) * 2, 0, 0);
            }
        }
    }
}
}  // namespace

// dh fp32 [rows][256] (gradient of h = x + o_proj(attn_out)), attn_out fp32 [rows][256] (H = 8 heads of 32), packed = the block image of
// gaot_block_pack_multi -> do_image bf16 [rows][256] and delta fp32 [rows / S][8][S] for gaot_attn_bwd_bf16 (phases 16 | 32 without 1 / 8)
extern "C" int gaot_oproj_bwd_image(const float* dh, const float* attn_out, const void* packed, int F, void* do_image, float* delta,
                                    int64_t rows, int S, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(dh && attn_out && packed && do_image && delta && rows > 0 && S > 0 && rows % S == 0 && F > 0 && F % FC == 0, "bad argument");
    GAOT_CHECK_ARG(((uintptr_t)dh % 16) == 0 && ((uintptr_t)attn_out % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)do_image % 16) == 0,
                   "16-byte alignment");
    if (rows * (int64_t)D * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_oproj_bwd_image: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    const bf16_t* wotp = (const bf16_t*)packed + ((int64_t)2 * F * D + (int64_t)D * F) * 2 + (int64_t)D * D;
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_oproj_bwd, dim3((unsigned)(8 * per)), dim3(256), H_BYTES, (hipStream_t)stream, dh, attn_out, (const u32x4*)wotp, (bf16_t*)do_image,
                 delta, (int)rows, S, D / 32);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// ---- the head of a Transformer block in one launch (reference attn.py:104-109, 118-120, 226: q / k / v projections of attn_norm(x), rotary
// embedding of q and k): x fp32 [rows][ldx] -> the attention kernels' bf16 image [rows][N] (N = (H + 2 HKV) * 32, a multiple of 256; RoPE
// from the [S][16] (cos, sin) table, q pre-scaled), yb = bf16(attn_norm(x)) (the weight-gradient product's operand) and rstd.  Stands in
// for gaot_rmsnorm_fwd + gaot_qkv_image (a read and two writes of [rows][256] less, one launch instead of two).  Row blocks of 64; the
// weights (N x 256) stream from L2 as fragment blocks (((p*4 + w)*2 + jt)*16 + s: row 256 p + 64 w + 32 jt + l31, k = 16 s + 8 hf + e):
// panel p = 256 output columns, wave w its heads 8 p + 2 w, + 1.  Arithmetic: the norm as in k_rmsnorm_fwd, the product and the epilogue as
// in k_gemm_k256<OUT_QKV_IMAGE> -- same values.
namespace {
struct QkvTable {
    const float* w[PACK_MAX];
    bf16_t* packed[PACK_MAX];
};
// blockIdx.y = the block; N x 256 weights -> fragment image (and, with_backward, the image of the transpose for the input gradient:
// block ((w*2 + jt)*(N/16) + s): "row" j = 64 w + 32 jt + l31 (the norm's column), k = 16 s + 8 hf + e over the N output columns -> w[k][j])
__global__ void k_qkv_pack(QkvTable t, int N, int with_backward) {
    const float* w = t.w[blockIdx.y];
    bf16_t* p = t.packed[blockIdx.y];
    const int nfr = N * D / 8;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < nfr * (with_backward ? 2 : 1); id += gridDim.x * blockDim.x) {
        if (id < nfr) {
            const int lane = id & 63, s2 = (id >> 6) & 15, jt = (id >> 10) & 1, wv = (id >> 11) & 3, pn = id >> 13;
            const float* src = w + (int64_t)(256 * pn + 64 * wv + 32 * jt + (lane & 31)) * D + 16 * s2 + 8 * (lane >> 5);
            *reinterpret_cast<u32x4*>(p + (int64_t)id * 8) = u32x4{pack2(src[0], src[1]), pack2(src[2], src[3]), pack2(src[4], src[5]), pack2(src[6], src[7])};
        } else {
            const int i2 = id - nfr, ns = N / 16;
            const int lane = i2 & 63, rest = i2 >> 6, s2 = rest % ns, jt = (rest / ns) & 1, wv = rest / (2 * ns);
            const float* st = w + (int64_t)(16 * s2 + 8 * (lane >> 5)) * D + (wv * 64 + jt * 32 + (lane & 31));
            *reinterpret_cast<u32x4*>(p + (int64_t)id * 8) =
                u32x4{pack2(st[0], st[D]), pack2(st[2 * D], st[3 * D]), pack2(st[4 * D], st[5 * D]), pack2(st[6 * D], st[7 * D])};
        }
    }
}

// blockIdx.y = the block; skip_proj.weight [256][512] -> blocks ((w*2 + jt)*32 + s): row 64 w + 32 jt + l31, k = 16 s + 8 hf + e
__global__ void k_skip_pack(QkvTable t) {
    const float* w = t.w[blockIdx.y];
    bf16_t* p = t.packed[blockIdx.y];
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < D * 2 * D / 8; id += gridDim.x * blockDim.x) {
        const int lane = id & 63, s2 = (id >> 6) & 31, jt = (id >> 11) & 1, wv = id >> 12;
        const float* src = w + (int64_t)(64 * wv + 32 * jt + (lane & 31)) * (2 * D) + 16 * s2 + 8 * (lane >> 5);
        *reinterpret_cast<u32x4*>(p + (int64_t)id * 8) = u32x4{pack2(src[0], src[1]), pack2(src[2], src[3]), pack2(src[4], src[5]), pack2(src[6], src[7])};
        // the transposed image behind it: block ((half*4 + w)*2 + jt)*16 + s: "row" j = 256 half + 64 w + 32 jt + l31, k = 16 s + 8 hf + e -> w[k][j]
        const int s3 = (id >> 6) & 15, jt3 = (id >> 10) & 1, w3 = (id >> 11) & 3, half = id >> 13;
        const float* st = w + (int64_t)(16 * s3 + 8 * (lane >> 5)) * (2 * D) + (256 * half + 64 * w3 + 32 * jt3 + (lane & 31));
        *reinterpret_cast<u32x4*>(p + (int64_t)D * 2 * D + (int64_t)id * 8) =
            u32x4{pack2(st[0], st[2 * D]), pack2(st[4 * D], st[6 * D]), pack2(st[8 * D], st[10 * D]), pack2(st[12 * D], st[14 * D])};
    }
}

struct QkvArgs {
    const float* X; int ldx; const float* NW; float eps; const u32x4* Wp; bf16_t* IMG; bf16_t* YB; float* RSTD;
    const float* table; int S, nq, nk; float qscale; int M, N;
    // CAT: the decoder block's skip projection in front (reference attn.py:222-225: x = skip_proj(cat([x, skip]))): X, XB = the two inputs,
    // WSp = skip_proj.weight ([256][512]) as fragment blocks ((w*2 + jt)*32 + s), BS its bias, XO receives the projected rows (fp32
    // [M][256]: the block's residual and the norm's input for the backward)
    const float* XB; int ldxb; const u32x4* WSp; const float* BS; float* XO;
};
constexpr int QKV_LDS = H_BYTES + 256 + 64 * 144, QKV_LDS_CAT = 2 * H_BYTES + 256 + 1024 + 64 * 144;
template <bool CAT>
__global__ __launch_bounds__(256, 1) void k_norm_qkv(QkvArgs a) {
    constexpr int RD = 3, LA = 2;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int M = a.M, N = a.N, NP = N / 256;
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wp, 0, N * D * 2, 0x00020000);
    // step g = 4 p + st of this wave: the k-slice st (64 deep) of panel p, 8 blocks (tiles jt 0, 1 x 4 k-steps); past the last: zeros
    u32x4 wr[RD][8];
    auto wload = [&](u32x4 (&dst)[8], int g) {
        const int pn = g >> 2, st = g & 3;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, ((((pn * 4 + wv) * 2 + jt) * 16) + 4 * st + s2) * 1024, 0));
    };
    wload(wr[0], 0);
    wload(wr[1], 1);
    float* rstd_l = reinterpret_cast<float*>(lds + (CAT ? 2 * H_BYTES : H_BYTES));
    if constexpr (CAT) {
        const int sw0 = l31 & 15;
        const int64_t xab = (int64_t)M * a.ldx * 4, xbb = (int64_t)M * a.ldxb * 4, xob = (int64_t)M * D * 4, ybbytes = (int64_t)M * D * 2;
        const __amdgpu_buffer_rsrc_t xars = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(xab > 0x7fffffff ? 0x7fffffff : xab), 0x00020000);
        const __amdgpu_buffer_rsrc_t xbrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.XB, 0, (int)(xbb > 0x7fffffff ? 0x7fffffff : xbb), 0x00020000);
        const __amdgpu_buffer_rsrc_t xors = __builtin_amdgcn_make_buffer_rsrc((void*)a.XO, 0, (int)(xob > 0x7fffffff ? 0x7fffffff : xob), 0x00020000);
        const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.YB, 0, (int)(ybbytes > 0x7fffffff ? 0x7fffffff : ybbytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t wsrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.WSp, 0, D * 2 * D * 2, 0x00020000);
        u32x4 wo[2][8];
        auto woload = [&](u32x4 (&dst)[8], int st) {      // st 0..7: k-steps 4 st .. + 3 of 32
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrs, lane * 16, (((wv * 2 + jt) * 32) + 4 * st + s2) * 1024, 0));
        };
        woload(wo[0], 0);
        woload(wo[1], 1);
        // the two inputs fp32 -> bf16 -> LDS tiles (x at 0, skip at H_BYTES): row 8 i + tid >> 5, 16-byte bf16 chunk tid & 31 = 8 floats
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 d0[8], d1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
                const unsigned off = m < M ? (unsigned)m * (unsigned)(half ? a.ldxb : a.ldx) * 4u + (unsigned)(threadIdx.x & 31) * 32u : 0x80000000u;
                d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(half ? xbrs : xars, off, 0, 0));
                d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(half ? xbrs : xars, off, 16, 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31;
                const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
                *reinterpret_cast<u32x4*>(lds + half * H_BYTES + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
            }
        }
        // the projected rows start as the bias: column 64 wave + 32 jt + mfma32_row(r, hf), row m0 + 32 i + l31
        f32x16 hacc[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = a.BS ? *reinterpret_cast<const f32x4*>(a.BS + wave * 64 + 32 * jt + 8 * q + 4 * hf) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    hacc[jt][i][4 * q] = bv[0]; hacc[jt][i][4 * q + 1] = bv[1]; hacc[jt][i][4 * q + 2] = bv[2]; hacc[jt][i][4 * q + 3] = bv[3];
                }
            }
        __builtin_amdgcn_s_barrier();       // both bf16 tiles are complete
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const char* tb = lds + (st >> 2) * H_BYTES;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int slot = ((2 * (4 * (st & 3) + s2) + hf) ^ sw0) << 4;
                const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(tb + l31 * 512 + slot), o1 = *reinterpret_cast<const bf16x8*>(tb + (32 + l31) * 512 + slot);
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wo[st & 1][s2]), w1 = __builtin_bit_cast(bf16x8, wo[st & 1][4 + s2]);
                hacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, o0, hacc[0][0], 0, 0, 0);
                hacc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, o0, hacc[1][0], 0, 0, 0);
                hacc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, o1, hacc[0][1], 0, 0, 0);
                hacc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, o1, hacc[1][1], 0, 0, 0);
            }
            if (st + 2 < 8) woload(wo[st & 1], st + 2);
        }
        float* part = rstd_l + 64;       // [4 waves][64 rows]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * i + l31;
            const unsigned rowoff = m < M ? (unsigned)m * (unsigned)D * 4u : 0x80000000u;
            float ssl = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {hacc[jt][i][4 * q], hacc[jt][i][4 * q + 1], hacc[jt][i][4 * q + 2], hacc[jt][i][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xors, rowoff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0);
                    ssl += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                }
            ssl += __shfl_xor(ssl, 32, 64);
            if (hf == 0) part[wave * 64 + 32 * i + l31] = ssl;
        }
        __builtin_amdgcn_s_barrier();       // partials complete; every wave is done reading the input tiles
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = 32 * i + l31, m = m0 + ml;
            const float ssr = ((part[ml] + part[64 + ml]) + part[128 + ml]) + part[192 + ml];
            const float r = rsqrtf(ssr / (float)D + a.eps);
            if (wave == 0 && hf == 0) {
                rstd_l[ml] = r;
                if (m < M) a.RSTD[m] = r;
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(a.NW + wave * 64 + 32 * jt + 8 * q + 4 * hf);
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 pk = {pack2(hacc[jt][i][4 * q] * r * g[0], hacc[jt][i][4 * q + 1] * r * g[1]),
                                      pack2(hacc[jt][i][4 * q + 2] * r * g[2], hacc[jt][i][4 * q + 3] * r * g[3])};
                    *reinterpret_cast<u32x2*>(lds + ml * 512 + (((8 * wave + 4 * jt + q) ^ (ml & 15)) << 4) + 8 * hf) = pk;
                }
        }
        __builtin_amdgcn_s_barrier();       // the normalised tile is complete
#pragma unroll
        for (int k = 0; k < 8; ++k) {        // bf16(norm(x')) to HBM, this wave's 16 rows, two rows per instruction
            const int row = 16 * wave + 2 * k + hf, ch = l31, m = m0 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + row * 512 + ((ch ^ (row & 15)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(v, ybrs, m < M ? (unsigned)m * (unsigned)D * 2u + (unsigned)ch * 16u : 0x80000000u, 0, 0);
        }
    } else {   // attn_norm: rows 16 wave .. + 15, a lane per float4 (the arithmetic and summation order of k_rmsnorm_fwd, rowops.hip)
        const int64_t xbytes = (int64_t)M * a.ldx * 4, ybbytes = (int64_t)M * D * 2;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.YB, 0, (int)(ybbytes > 0x7fffffff ? 0x7fffffff : ybbytes), 0x00020000);
        const float4 g = reinterpret_cast<const float4*>(a.NW)[lane];
        float4 v[16];
        float ss[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = m0 + wave * 16 + j;
            v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, m < M ? (unsigned)m * (unsigned)a.ldx * 4u + lane * 16u : 0x80000000u, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            ss[j] = 0.f;
            ss[j] += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 16; ++j) ss[j] += __shfl_xor(ss[j], o, 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rl = wave * 16 + j, m = m0 + rl;
            const float r = rsqrtf(ss[j] / (float)D + a.eps);
            const float4 o = make_float4(v[j].x * r * g.x, v[j].y * r * g.y, v[j].z * r * g.z, v[j].w * r * g.w);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 pk = {pack2(o.x, o.y), pack2(o.z, o.w)};
            *reinterpret_cast<u32x2*>(lds + rl * 512 + ((((lane >> 1) ^ (rl & 15))) << 4) + 8 * (lane & 1)) = pk;
            __builtin_amdgcn_raw_buffer_store_b64(pk, ybrs, m < M ? (unsigned)m * (unsigned)D * 2u + lane * 8u : 0x80000000u, 0, 0);
            if (lane == 0) {
                rstd_l[rl] = r;
                if (m < M) a.RSTD[m] = r;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    const int64_t ibytes = (int64_t)M * N * 2;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)a.IMG, 0, (int)(ibytes > 0x7fffffff ? 0x7fffffff : ibytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.table, 0, a.table ? a.S * 128 : 0, 0x00020000);
    const int sw = l31 & 15;
    const char* hb = lds + l31 * 512;
    // the (cos, sin) rows of the block's 64 positions (128 B each; the same for every head) are fetched ONCE into LDS (pitch 144 B: the
    // 16-byte reads of 16 consecutive rows cover all banks) -- in the panel epilogues every one of these loads sat in front of its use,
    // an exposed L2 round trip per batch; held in registers they cost the 32 that made the CAT form spill
    char* tabl = lds + (CAT ? QKV_LDS_CAT : QKV_LDS) - 64 * 144;
    {
        const int row = threadIdx.x >> 2, pc = threadIdx.x & 3, m = m0 + row, mm = m < M ? m : M - 1;
        const f32x4 t0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, (mm % a.S) * 128 + 32 * pc, 0, 0));
        const f32x4 t1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, (mm % a.S) * 128 + 32 * pc + 16, 0, 0));
        *reinterpret_cast<f32x4*>(tabl + row * 144 + 32 * pc) = t0;
        *reinterpret_cast<f32x4*>(tabl + row * 144 + 32 * pc + 16) = t1;
    }
    __builtin_amdgcn_s_barrier();
    // the product of panel pn + 1 (MFMAs) carries the epilogue of panel pn (vector instructions, stores) between its k-steps: two
    // accumulator sets; a panel past the last reads zero weights (buffer range check) -- one product too many instead of a second copy
    // of the loop body
    f32x16 accc[2][2], accn[2][2];
    bf16x8 hn0 = *reinterpret_cast<const bf16x8*>(hb + ((hf ^ sw) << 4)), hn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + ((hf ^ sw) << 4));
    // one 64-deep step (four k-steps) of panel pn into acc; epi(k-step) after each k-step's MFMAs
    auto panel_step = [&](f32x16 (&acc)[2][2], int pn, int st, auto epi) {
        const int g = 4 * pn + st;
        auto run = [&](auto rot) {
            constexpr int R = decltype(rot)::value;        // R = (4 pn) % 3: the ring slot of step g is (R + st) % 3
            wload(wr[(R + st + LA) % RD], g + LA);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 h0 = hn0, h1 = hn1;
                const int slot = ((2 * ((4 * st + s2 + 1) & 15) + hf) ^ sw) << 4;
                hn0 = *reinterpret_cast<const bf16x8*>(hb + slot);
                hn1 = *reinterpret_cast<const bf16x8*>(hb + 32 * 512 + slot);
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wr[(R + st) % RD][s2]), w1 = __builtin_bit_cast(bf16x8, wr[(R + st) % RD][4 + s2]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, h0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, h0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, h1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, h1, acc[1][1], 0, 0, 0);
                epi(4 * st + s2);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
            }
        };
        const int rot = (4 * pn) % 3;
        if (rot == 0) run(std::integral_constant<int, 0>{});
        else if (rot == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
    };
    auto zero4 = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
    };
    zero4(accc);
#pragma unroll
    for (int st = 0; st < 4; ++st) panel_step(accc, 0, st, [](int) {});
    for (int pn = 0; pn < NP; ++pn) {
        // epilogue of panel pn in eight pieces (k-steps 0, 2, .., 14 of the next panel's product): piece e = (row tile i = e >> 2, head tile
        // jt = (e >> 1) & 1, run pair q = e & 1 with q + 2).  accc[jt][i][r] = head 8 pn + 2 wave + jt, its column mfma32_row(r, hf), row
        // m0 + 32 i + l31; the lane holds runs of 4 consecutive columns 8q + 4hf .. +3 = two rotation pairs (frequencies 4q + 2hf, + 1)
        auto piece = [&](int e) {
            const int i = e >> 2, jt = (e >> 1) & 1, q0 = e & 1;
            const int m = m0 + 32 * i + l31;
            const unsigned rowoff = m < M ? (unsigned)m * (unsigned)N * 2u : 0x80000000u;
            const int head = 8 * pn + 2 * wave + jt;
            const bool rope = a.table && head < a.nq + a.nk;
            const float sc = head < a.nq ? a.qscale : 1.0f;
            unsigned pk[2][2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int q = q0 + 2 * h2;
                float v0 = accc[jt][i][4 * q], v1 = accc[jt][i][4 * q + 1], v2 = accc[jt][i][4 * q + 2], v3 = accc[jt][i][4 * q + 3];
                if (rope) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(tabl + (32 * i + l31) * 144 + 16 * hf + 32 * q);   // cos, sin, cos, sin
                    const float r0 = v0 * tb[0] - v1 * tb[1], r1 = v1 * tb[0] + v0 * tb[1];
                    const float r2 = v2 * tb[2] - v3 * tb[3], r3 = v3 * tb[2] + v2 * tb[3];
                    v0 = r0; v1 = r1; v2 = r2; v3 = r3;
                }
                pk[h2][0] = pack2(v0 * sc, v1 * sc);
                pk[h2][1] = pack2(v2 * sc, v3 * sc);
            }
            // runs (q, q+2): lower lanes end with columns 8q..8q+7, upper lanes with 16+8q..
            const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(v, irs, rowoff + (head * 32 + 8 * q0 + 16 * hf) * 2, 0, 0);
        };
        zero4(accn);
#pragma unroll
        for (int st = 0; st < 4; ++st)
            panel_step(accn, pn + 1, st, [&](int ks) { if ((ks & 1) == 0) piece(ks >> 1); });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 2; ++i) accc[jt][i] = accn[jt][i];
    }
}
}  // namespace

extern "C" int64_t gaot_qkv_packed_bytes(int64_t N, int with_backward) { return N * D * 2 * (with_backward ? 2 : 1); }

// fp32 co-located q | k | v weights ([N][256], N = (H + 2 HKV) * 32, a multiple of 256) of every block -> fragment images, one launch
extern "C" int gaot_qkv_pack_multi(const gaot_qkv_pack_t* items, int num, int64_t N, int with_backward, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(items && num > 0 && N > 0 && N % 256 == 0, "bad argument (N must be a multiple of 256)");
    for (int i0 = 0; i0 < num; i0 += PACK_MAX) {
        QkvTable t{};
        const int n = std::min(PACK_MAX, num - i0);
        for (int i = 0; i < n; ++i) {
            GAOT_CHECK_ARG(items[i0 + i].w && items[i0 + i].packed && ((uintptr_t)items[i0 + i].packed % 16) == 0, "null or misaligned pointer in the table");
            t.w[i] = items[i0 + i].w; t.packed[i] = (bf16_t*)items[i0 + i].packed;
        }
        GAOT_KLAUNCH(k_qkv_pack, dim3((unsigned)std::min<int64_t>(ceil_div(N * D / 8 * (with_backward ? 2 : 1), 256), 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, t, (int)N, with_backward);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_norm_qkv_image(const float* x, int64_t ldx, const float* norm_weight, float eps, const void* packed, void* image, void* yb,
                                   float* rstd, int64_t rows, int S, int H, int HKV, const float* rope_table, float qscale, gaot_stream_t stream) {
    GAOT_ENTER();
    const int64_t N = (int64_t)(H + 2 * HKV) * 32;
    GAOT_CHECK_ARG(x && norm_weight && packed && image && yb && rstd && rows > 0 && S > 0 && H > 0 && HKV > 0 && N % 256 == 0,
                   "bad argument ((H + 2 HKV) * 32 must be a multiple of 256)");
    GAOT_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)norm_weight % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)image % 16) == 0 &&
                   ((uintptr_t)yb % 16) == 0 && ldx % 4 == 0 && ldx >= D, "16-byte alignment");
    if (rows * N * 2 >= 0x7fffffff || rows * ldx * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_norm_qkv_image: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_norm_qkv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, QKV_LDS);
        if (e != hipSuccess) {
            gaot_set_error("norm_qkv: cannot set dynamic LDS %d: %s", QKV_LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const QkvArgs a{x, (int)ldx, norm_weight, eps, (const u32x4*)packed, (bf16_t*)image, (bf16_t*)yb, rstd, rope_table, S, H, HKV, qscale, (int)rows, (int)N,
                    nullptr, 0, nullptr, nullptr, nullptr};
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_norm_qkv<false>, dim3((unsigned)(8 * per)), dim3(256), QKV_LDS, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// the same with the decoder block's skip projection in front (reference attn.py:222-225: x = skip_proj(cat([x, skip]))): xa, xb fp32
// [rows][256] (ldxa, ldxb), skip_packed = gaot_skip_pack_multi's image of skip_proj.weight ([256][512]), skip_bias [256] or NULL;
// x_out receives the projected rows (fp32 [rows][256]).  Stands in for two gaot_gemm launches + gaot_rmsnorm_fwd + gaot_qkv_image.
extern "C" int64_t gaot_skip_packed_bytes(void) { return 2 * (int64_t)D * 2 * D * 2; }      // forward image + transposed image
extern "C" int gaot_skip_pack_multi(const gaot_qkv_pack_t* items, int num, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(items && num > 0, "bad argument");
    for (int i0 = 0; i0 < num; i0 += PACK_MAX) {
        QkvTable t{};
        const int n = std::min(PACK_MAX, num - i0);
        for (int i = 0; i < n; ++i) {
            GAOT_CHECK_ARG(items[i0 + i].w && items[i0 + i].packed && ((uintptr_t)items[i0 + i].packed % 16) == 0, "null or misaligned pointer in the table");
            t.w[i] = items[i0 + i].w; t.packed[i] = (bf16_t*)items[i0 + i].packed;
        }
        GAOT_KLAUNCH(k_skip_pack, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
extern "C" int gaot_cat_norm_qkv_image(const float* xa, int64_t ldxa, const float* xb, int64_t ldxb, const void* skip_packed, const float* skip_bias,
                                       float* x_out, const float* norm_weight, float eps, const void* packed, void* image, void* yb, float* rstd,
                                       int64_t rows, int S, int H, int HKV, const float* rope_table, float qscale, gaot_stream_t stream) {
    GAOT_ENTER();
    const int64_t N = (int64_t)(H + 2 * HKV) * 32;
    GAOT_CHECK_ARG(xa && xb && skip_packed && x_out && norm_weight && packed && image && yb && rstd && rows > 0 && S > 0 && H > 0 && HKV > 0 &&
                   N % 256 == 0, "bad argument ((H + 2 HKV) * 32 must be a multiple of 256)");
    GAOT_CHECK_ARG(((uintptr_t)xa % 16) == 0 && ((uintptr_t)xb % 16) == 0 && ((uintptr_t)skip_packed % 16) == 0 && ((uintptr_t)skip_bias % 16) == 0 &&
                   ((uintptr_t)x_out % 16) == 0 && ((uintptr_t)norm_weight % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)image % 16) == 0 &&
                   ((uintptr_t)yb % 16) == 0 && ldxa % 4 == 0 && ldxa >= D && ldxb % 4 == 0 && ldxb >= D, "16-byte alignment");
    if (rows * N * 2 >= 0x7fffffff || rows * ldxa * 4 >= 0x7fffffff || rows * ldxb * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_cat_norm_qkv_image: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_norm_qkv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, QKV_LDS_CAT);
        if (e != hipSuccess) {
            gaot_set_error("cat_norm_qkv: cannot set dynamic LDS %d: %s", QKV_LDS_CAT, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const QkvArgs a{xa, (int)ldxa, norm_weight, eps, (const u32x4*)packed, (bf16_t*)image, (bf16_t*)yb, rstd, rope_table, S, H, HKV, qscale, (int)rows, (int)N,
                    xb, (int)ldxb, (const u32x4*)skip_packed, skip_bias, x_out};
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_norm_qkv<true>, dim3((unsigned)(8 * per)), dim3(256), QKV_LDS_CAT, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// gaot_ffn_bwd with ffn_norm's backward in its epilogue (reference attn.py:227-229 autograd): yb = bf16(ffn_norm(h)) as saved by
// gaot_norm_ffn_fwd / gaot_block_tail_fwd, h fp32 [rows][ldh], rstd; the residual's gradient dy is part of d(norm) (add_dy = 1 always).
// -> dh fp32 [rows][256], dag, u, dyb as gaot_ffn_bwd, and dw_part fp32 [ceil(rows / 64)][256]: the norm weight's gradient is the sum of
// its rows (gaot_reduce_multi, parts = gaot_norm_bwd_parts(rows)).  Stands in for gaot_ffn_bwd + gaot_rmsnorm_bwd.
extern "C" int64_t gaot_norm_bwd_parts(int64_t rows) { return (rows + RB - 1) / RB; }

extern "C" int gaot_ffn_bwd_norm(const void* yb, const float* dy, const void* packed, const float* h, int64_t ldh, const float* norm_weight,
                                 const float* rstd, void* dag, void* u, void* dyb, float* dh, float* dw_part, int64_t rows, int F,
                                 gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(yb && dy && packed && h && norm_weight && rstd && dag && u && dh && dw_part && rows > 0 && F > 0 && F % FC == 0,
                   "bad argument (F must be a multiple of 128)");
    GAOT_CHECK_ARG(((uintptr_t)yb % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dag % 16) == 0 &&
                   ((uintptr_t)u % 16) == 0 && ((uintptr_t)dyb % 16) == 0 && ((uintptr_t)dh % 16) == 0 && ((uintptr_t)h % 16) == 0 &&
                   ((uintptr_t)norm_weight % 16) == 0 && ldh % 4 == 0 && ldh >= D, "16-byte alignment");
    if (rows * (int64_t)F * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_ffn_bwd_norm: rows * F too large for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static_assert(BWDX_LDS >= NB_LDS, "the epilogue reuses the kernel's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_bwd_dx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWDX_LDS);
        if (e != hipSuccess) {
            gaot_set_error("ffn_bwd_norm: cannot set dynamic LDS %d: %s", BWDX_LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const bf16_t* p = (const bf16_t*)packed;
    const bf16_t* w2tp = p + (int64_t)2 * F * D + (int64_t)D * F;
    const bf16_t* w13tp = w2tp + (int64_t)D * F;
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_ffn_bwd_dx<true>, dim3((unsigned)(8 * per)), dim3(256), BWDX_LDS, (hipStream_t)stream, (const bf16_t*)yb, dy, (const u32x4*)p,
                 (const u32x4*)w2tp, (const u32x4*)w13tp, (bf16_t*)dag, (bf16_t*)u, (bf16_t*)dyb, dh, (int)rows, F, 1,
                 NormBwdArgs{h, (int)ldh, norm_weight, rstd, dw_part});
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// ---- the q | k | v projections' input gradient with attn_norm's backward in its epilogue (reference attn.py:104-106, 226 autograd):
// dqkv fp32 [rows][N] (N a multiple of 256), packed = gaot_qkv_pack_multi's image WITH the backward image, x fp32 [rows][ldx] (the
// block's input), rstd; dres / dtap (may be NULL): the gradients reaching x through the block's residual and through the long-range
// skip tap -> dx fp32 [rows][256], dw_part [ceil(rows / 64)][256].  Stands in for gaot_gemm_ex (d(norm x) = dqkv Wqkv) + gaot_rmsnorm_bwd2.
namespace {
struct QkvBwdArgs {
    const float* DQKV; const u32x4* WTp; const float* X; int ldx; const float* NW; const float* RSTD; const float* DRES; const float* DTAP;
    float* DX; float* DWP; int M, N;
    // CATB: the decoder block's skip projection behind x (x = skip_proj(cat([xa, xb])), attn.py:222-225): its two input gradients
    // dxa = dx Ws[:, :256], dxb = dx Ws[:, 256:] from the dx rows on chip (bf16 tile); WSTp = fragments of Ws^T: blocks
    // ((half*4 + w)*2 + jt)*16 + s: "row" 256 half + 64 w + 32 jt + l31, k = 16 s + 8 hf + e -> ws[k][row]; same: xa is xb, one sum
    const u32x4* WSTp; float* DXA; float* DXB; int same;
};
constexpr int QKVB_LDS = NB_LDS > 2 * H_BYTES ? NB_LDS : 2 * H_BYTES, QKVB_LDS_CAT = NB_LDS + H_BYTES;
template <bool CATB>
__global__ __launch_bounds__(256, 1) void k_qkv_bwd_norm(QkvBwdArgs a) {
    constexpr int RD = 3, LA = 2;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int M = a.M, N = a.N, NS = N / 256;
    const int nblk = (M + RB - 1) / RB, per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * RB;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.WTp, 0, N * D * 2, 0x00020000);
    const int64_t gbytes = (int64_t)M * N * 4;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)a.DQKV, 0, (int)(gbytes > 0x7fffffff ? 0x7fffffff : gbytes), 0x00020000);
    // step g = 4 sl + st: k-steps 16 sl + 4 st .. + 3 of this wave's two column tiles: blocks ((w*2 + jt)*(N/16) + k-step)
    const int ns16 = N / 16;
    u32x4 wr[RD][8];
    auto wload = [&](u32x4 (&dst)[8], int g) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, (((wv * 2 + jt) * ns16) + 4 * g + s2) * 1024, 0));
    };
    wload(wr[0], 0);
    wload(wr[1], 1);
    // K-slice sl of the dqkv rows (256 columns) fp32 -> registers (row 8 i + tid >> 5, 8 floats at column 8 (tid & 31))
    f32x4 d0[8], d1[8];
    auto gload = [&](int sl) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), m = m0 + row;
            const unsigned off = m < M ? (unsigned)m * (unsigned)N * 4u + (unsigned)(sl * 256 + (threadIdx.x & 31) * 8) * 4u : 0x80000000u;
            d0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 0));
            d1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, off, 16, 0));
        }
    };
    auto gstore = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (threadIdx.x >> 5), ch = threadIdx.x & 31;
            const u32x4 v = {pack2(d0[i][0], d0[i][1]), pack2(d0[i][2], d0[i][3]), pack2(d1[i][0], d1[i][1]), pack2(d1[i][2], d1[i][3])};
            *reinterpret_cast<u32x4*>(buf + row * 512 + ((ch ^ (row & 15)) << 4)) = v;
        }
    };
    gload(0);
    gstore(lds);
    __builtin_amdgcn_s_barrier();
    f32x16 acc[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
    const int sw = l31 & 15;
    for (int sl = 0; sl < NS; ++sl) {
        const char* tb = lds + (sl & 1) * H_BYTES + l31 * 512;
        if (sl + 1 < NS) gload(sl + 1);
        auto run = [&](auto rot) {
            constexpr int R = decltype(rot)::value;        // (4 sl) % 3
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                wload(wr[(R + st + LA) % RD], 4 * sl + st + LA);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int slot = ((2 * (4 * st + s2) + hf) ^ sw) << 4;
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(tb + slot), a1 = *reinterpret_cast<const bf16x8*>(tb + 32 * 512 + slot);
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wr[(R + st) % RD][s2]), w1 = __builtin_bit_cast(bf16x8, wr[(R + st) % RD][4 + s2]);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, acc[1][1], 0, 0, 0);
                }
            }
        };
        const int rot = (4 * sl) % 3;
        if (rot == 0) run(std::integral_constant<int, 0>{});
        else if (rot == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
        if (sl + 1 < NS) gstore(lds + ((sl + 1) & 1) * H_BYTES);
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (!CATB) {
        norm_bwd_epilogue(acc, lds, a.X, a.ldx, a.NW, a.RSTD, a.DRES, a.DTAP, a.DX, a.DWP, M, m0, t);
    } else {
        char* dxt = lds + NB_LDS;
        norm_bwd_epilogue(acc, lds, a.X, a.ldx, a.NW, a.RSTD, a.DRES, a.DTAP, a.DX, a.DWP, M, m0, t, dxt);
        __builtin_amdgcn_s_barrier();       // the bf16 dx tile is complete
        int sw2 = l31 & 15;
        asm volatile("" : "+v"(sw2));       // keeps the tile addresses below from being formed (and held) at the top of the kernel
        const __amdgpu_buffer_rsrc_t wsrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.WSTp, 0, D * 2 * D * 2, 0x00020000);
        u32x4 wo[2][8];
        auto woload = [&](u32x4 (&dst)[8], int g) {      // g = 4 half + st: k-steps 4 st .. + 3 of the half's product
            const int half = g >> 2, st = g & 3;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    dst[jt * 4 + s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrs, lane * 16, (((((half * 4 + wv) * 2 + jt) * 16) + 4 * st + s2)) * 1024, 0));
        };
        woload(wo[0], 0);
        woload(wo[1], 1);
        f32x16 ga[2][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 0 || !a.same) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ga[jt][i][r] = 0.f;
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int g = 4 * half + st;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int slot = ((2 * (4 * st + s2) + hf) ^ sw2) << 4;
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(dxt + l31 * 512 + slot), a1 = *reinterpret_cast<const bf16x8*>(dxt + (32 + l31) * 512 + slot);
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wo[g & 1][s2]), w1 = __builtin_bit_cast(bf16x8, wo[g & 1][4 + s2]);
                    ga[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, ga[0][0], 0, 0, 0);
                    ga[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, ga[1][0], 0, 0, 0);
                    ga[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a1, ga[0][1], 0, 0, 0);
                    ga[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, ga[1][1], 0, 0, 0);
                }
                if (g + 2 < 8) woload(wo[g & 1], g + 2);
            }
            if (half == 1 || !a.same) {       // same: both halves' products accumulate into ONE gradient, written after the second
                float* out = (half == 0 || a.same) ? a.DXA : a.DXB;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + 32 * i + l31;
                    if (m < M) {
#pragma unroll
                        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *reinterpret_cast<f32x4*>(out + (int64_t)m * D + wave * 64 + 32 * jt + 8 * q + 4 * hf) =
                                    f32x4{ga[jt][i][4 * q], ga[jt][i][4 * q + 1], ga[jt][i][4 * q + 2], ga[jt][i][4 * q + 3]};
                    }
                }
            }
        }
    }
}
}  // namespace

extern "C" int gaot_qkv_bwd_norm(const float* dqkv, int64_t N, const void* packed, const float* x, int64_t ldx, const float* norm_weight,
                                 const float* rstd, const float* dres, const float* dtap, float* dx, float* dw_part, int64_t rows,
                                 gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(dqkv && packed && x && norm_weight && rstd && dx && dw_part && rows > 0 && N > 0 && N % 256 == 0, "bad argument (N must be a multiple of 256)");
    GAOT_CHECK_ARG(((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)norm_weight % 16) == 0 &&
                   ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dtap % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ldx % 4 == 0 && ldx >= D, "16-byte alignment");
    if (rows * N * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_qkv_bwd_norm: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_qkv_bwd_norm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, QKVB_LDS);
        if (e != hipSuccess) {
            gaot_set_error("qkv_bwd_norm: cannot set dynamic LDS %d: %s", QKVB_LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const QkvBwdArgs a{dqkv, (const u32x4*)((const bf16_t*)packed + N * D), x, (int)ldx, norm_weight, rstd, dres, dtap, dx, dw_part, (int)rows, (int)N,
                       nullptr, nullptr, nullptr, 0};
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_qkv_bwd_norm<false>, dim3((unsigned)(8 * per)), dim3(256), QKVB_LDS, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// the same for a decoder block (x = skip_proj(cat([xa, xb]))): also the skip projection's two input gradients dxa = dx Ws[:, :256],
// dxb = dx Ws[:, 256:] (fp32 [rows][256] each; same != 0: xa is xb -- one gradient, the sum, in dxa; dxb unused) from the dx rows on chip.
// skip_packed: gaot_skip_pack_multi's image WITH the transposed image behind it (gaot_skip_packed_bytes() covers both).
extern "C" int gaot_qkv_bwd_norm_cat(const float* dqkv, int64_t N, const void* packed, const float* x, int64_t ldx, const float* norm_weight,
                                     const float* rstd, const float* dres, const void* skip_packed, float* dx, float* dxa, float* dxb, int same,
                                     float* dw_part, int64_t rows, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(dqkv && packed && x && norm_weight && rstd && skip_packed && dx && dxa && (dxb || same) && dw_part && rows > 0 && N > 0 && N % 256 == 0,
                   "bad argument (N must be a multiple of 256)");
    GAOT_CHECK_ARG(((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)norm_weight % 16) == 0 &&
                   ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)dxa % 16) == 0 && ((uintptr_t)dxb % 16) == 0 &&
                   ((uintptr_t)skip_packed % 16) == 0 && ldx % 4 == 0 && ldx >= D, "16-byte alignment");
    if (rows * N * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_qkv_bwd_norm_cat: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_qkv_bwd_norm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, QKVB_LDS_CAT);
        if (e != hipSuccess) {
            gaot_set_error("qkv_bwd_norm_cat: cannot set dynamic LDS %d: %s", QKVB_LDS_CAT, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const QkvBwdArgs a{dqkv, (const u32x4*)((const bf16_t*)packed + N * D), x, (int)ldx, norm_weight, rstd, dres, nullptr, dx, dw_part, (int)rows, (int)N,
                       (const u32x4*)((const bf16_t*)skip_packed + (int64_t)D * 2 * D), dxa, dxb, same};
    const int nblk = ((int)rows + RB - 1) / RB, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(k_qkv_bwd_norm<true>, dim3((unsigned)(8 * per)), dim3(256), QKVB_LDS_CAT, (hipStream_t)stream, a);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
