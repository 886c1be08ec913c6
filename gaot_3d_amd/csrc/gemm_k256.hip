// C[M,N] = A[M,256] . W[N,256]^T for the d_model = 256 projections of the latent Transformer whose operands already are
// bf16 in memory (q|k|v and w1|w3 forward: x is the bf16 image RMSNorm wrote, W the per-forward bf16 weight copy;
// reference attn.py:104-106,156 -> nn.Linear).  These GEMMs are output-write / latency bound (K = 256: four 64-deep
// steps), so the design removes re-reads and staging work instead of chasing MFMA issue:
//   * WEIGHTS IN REGISTERS: a wave keeps its 64 output columns x K = 256 of W as 128 VGPRs of ready MFMA fragments for the
//     whole launch (the generic kernel re-reads the weight panel from L2 once per 64-row tile: 256 times at M = 16 384);
//   * activation rows arrive as 64-row blocks by LDS-DMA (buffer_load ... lds, 1 KB per wave-instruction, no VGPR staging);
//     the LDS image is lane-linear, the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied
//     to the SOURCE address; two blocks in LDS, the next block's DMA is waited for AFTER this block's MFMAs;
//   * the product is formed transposed (W fragment as the MFMA's A operand), so a lane owns one output ROW and stores
//     16 bytes at a time (bf16 results: v_permlane32_swap pairs the two half-waves' 8-byte runs);
//   * workgroup id -> (XCD, column panel, row chunk): the workgroups of one XCD share one 1/8 slice of the rows.
// Measured (tools/lab/gemm_k256_lab.hip, M = 16 384): N = 2048 bf16 out 53.8 -> 32.0 us, N = 768 fp32 out 25.4 -> 22.5 us.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KK = 256, RB = 64, STAGE = RB * KK * 2;   // one staged row block: 64 rows x 512 B = 32 KB

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(bf16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)b) << 16);
}

// what the epilogue writes: fp32 C, bf16 C, or the attention kernels' bf16 image of a fused q|k|v projection (RoPE applied to
// the q and k heads from a [S][16] (cos, sin) table, q pre-scaled): the fp32 projection then never exists in HBM
// OUT_SWIGLU_BWD (gaot_ffn_w2_bwd_swiglu): the product is du = dy W2 (N = F columns); the epilogue reads a | g (bf16 [M][2F]) at the
// tile's own positions and writes da | dg (bf16 [M][2F]) -- du itself never exists in HBM and the stand-alone SwiGLU backward pass
// (a read of a | g, a read of du, a write of da | dg) is gone
enum { OUT_F32 = 0, OUT_BF16 = 1, OUT_QKV_IMAGE = 2, OUT_SWIGLU = 3, OUT_SWIGLU_BWD = 4 };
struct ImageArgs {
    const float* table;   // [S][16][2] = (cos, sin) of position * frequency, or null (no RoPE)
    int S, nq, nk;        // rows per sequence, number of q heads, of k heads (32 columns each; the rest are v heads)
    float qscale;
    // pack_g > 1 (sequence-parallel exchange, gaot_qkv_image_packed): the image is written as pack_g blocks [M][lw], block j =
    // q | k | v of rank j's heads (lw = (nq + 2 nk) / pack_g * 32) -- the send buffer of the all-to-all, no repacking pass
    int pack_g;
    // OUT_SWIGLU (gaot_ffn_w13_swiglu): W = [w1; w3] ([2F][256]); a wave's two 32-column tiles are columns n .. n+31 of w1 x and the
    // SAME columns of w3 x, so the epilogue writes a | g (bf16 [M][2F], kept for the backward) and u = silu(a) g (bf16 [M][F])
    void* u;              // OUT_SWIGLU_BWD: the a | g input (read only)
    int F;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_gemm_k256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, void* __restrict__ C,
                                                       int M, int N, int lda, int ldw, int ldc, int P, int subs, ImageArgs im) {
    constexpr bool C16 = MODE != OUT_F32;   // OUT_SWIGLU: a | g are written as bf16 like OUT_BF16
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, panel = j % P, sub = j / P;
    const int nblk = (M + RB - 1) / RB, c = xcd * subs + sub, nch = 8 * subs;
    const int t0 = (int)((int64_t)c * nblk / nch), t1 = (int)((int64_t)(c + 1) * nblk / nch);
    if (t0 >= t1) return;                                    // uniform over the workgroup
    // OUT_SWIGLU: a panel is 128 columns of a and the same 128 columns of g (tile 0 / tile 1 of a wave: columns n0 .. n0+31 of each)
    const int n0 = MODE == OUT_SWIGLU ? panel * 128 + wave * 32 : panel * 256 + wave * 64;
    const bool wave_ok = MODE == OUT_SWIGLU ? n0 < im.F : n0 < N;   // N % 64 == 0 (F % 32 == 0): all inside or all outside

    // the wave's weight slice as MFMA fragments: element (n = l31, k = 16 s + 8 hf + 0..7) of column tile jt
    // OUT_SWIGLU_BWD's epilogue needs 8 registers more than there are beside 128 of weights and 64 of accumulators: the last RELOAD
    // weight fragments of column tile 1 are not kept across the epilogue but fetched again (L2 hits, 16 bytes per lane each) at the
    // top of every row block, long before the k-steps that use them -- no scratch (a kernel with scratch pays ~5 us of idle queue
    // on either side of its launch)
    constexpr int RELOAD = MODE == OUT_SWIGLU_BWD ? 3 : 0;
    bf16x8 bw[2][16];
    const bf16_t* wrow1 = nullptr;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        int n = (MODE == OUT_SWIGLU ? n0 + jt * im.F : n0 + 32 * jt) + l31;
        n = n < N ? n : N - 1;
        const bf16_t* p = W + (int64_t)n * ldw + 8 * hf;
        if (jt == 1) wrow1 = p;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (!(jt == 1 && s >= 16 - RELOAD)) bw[jt][s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
    }

    const int64_t abytes = (int64_t)M * lda * 2, cbytes = (int64_t)M * ldc * (C16 ? 2 : 4);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(abytes > 0x7fffffff ? 0x7fffffff : abytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(cbytes > 0x7fffffff ? 0x7fffffff : cbytes), 0x00020000);
    const int64_t ubytes = MODE == OUT_SWIGLU ? (int64_t)M * im.F * 2 : (MODE == OUT_SWIGLU_BWD ? (int64_t)M * im.F * 4 : 0);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((MODE == OUT_SWIGLU || MODE == OUT_SWIGLU_BWD) ? im.u : C, 0, (int)(ubytes > 0x7fffffff ? 0x7fffffff : ubytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == OUT_QKV_IMAGE ? im.table : nullptr), 0,
                                                                         MODE == OUT_QKV_IMAGE && im.table ? im.S * 128 : 0, 0x00020000);
    auto stage = [&](int t, int buf) {
        const int lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // DMA piece q = wave*8 + i fills tile rows 2q, 2q+1: lane>>5 picks the row, lane&31 the 16-byte slot of the row;
            // slot s of row r holds source chunk s ^ (r & 15).  The whole row offset sits in voffset (soffset 0): the hardware
            // range check covers voffset only, so rows past M really read as zero instead of reading past the end of A
            const int x = 2 * i + lh;                                    // == row & 15
            const int voff = (t * RB + wave * 16 + x) * lda * 2 + (((lane & 31) ^ x) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (wave * 8 + i) * 1024), 16, voff,
                                                     0, 0, 0);
        }
    };

    stage(t0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        __builtin_amdgcn_s_barrier();            // block t has landed for every wave; nobody still reads the other buffer
        if (t + 1 < t1) stage(t + 1, buf ^ 1);
        if constexpr (RELOAD > 0) {
            const bf16_t* pw = wrow1;
            asm volatile("" : "+v"(pw));         // a new value every row block: the loads stay inside the loop
#pragma unroll
            for (int s = 16 - RELOAD; s < 16; ++s) bw[1][s] = *reinterpret_cast<const bf16x8*>(pw + 16 * s);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
        const char* base = lds + buf * STAGE + l31 * 512;
        const int sw = l31 & 15;
        bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base + ((hf ^ sw) << 4));
        bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + 32 * 512 + ((hf ^ sw) << 4));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8 b0 = a0, b1 = a1;
            if (s + 1 < 16) {
                a0 = *reinterpret_cast<const bf16x8*>(base + (((2 * s + 2 + hf) ^ sw) << 4));
                a1 = *reinterpret_cast<const bf16x8*>(base + 32 * 512 + (((2 * s + 2 + hf) ^ sw) << 4));
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][s], b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[1][s], b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][s], b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[1][s], b1, acc[1][1], 0, 0, 0);
        }
        // the next block's DMA (and the previous block's stores) have had the whole MFMA phase to complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // acc[jt][i][r]: column n = n0 + 32 jt + mfma32_row(r, hf), row m = 64 t + 32 i + l31
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = t * RB + 32 * i + l31;
            unsigned rowoff = (m < M && wave_ok) ? (unsigned)m * (unsigned)ldc * (C16 ? 2u : 4u) : 0x80000000u;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                int ncol = MODE == OUT_SWIGLU ? n0 + jt * im.F : n0 + 32 * jt;        // first output column of this 32-wide tile
                if constexpr (MODE == OUT_QKV_IMAGE) {
                    if (im.pack_g > 1) {        // uniform: destination = block of the rank that owns this head
                        const int head = ncol >> 5, hl = im.nq / im.pack_g, kl = im.nk / im.pack_g, lw = (hl + 2 * kl) * 32;
                        int j, within;
                        if (head < im.nq) { j = head / hl; within = (head % hl) * 32; }
                        else if (head < im.nq + im.nk) { j = (head - im.nq) / kl; within = (hl + (head - im.nq) % kl) * 32; }
                        else { j = (head - im.nq - im.nk) / kl; within = (hl + kl + (head - im.nq - im.nk) % kl) * 32; }
                        rowoff = (m < M && wave_ok) ? ((unsigned)j * (unsigned)M + (unsigned)m) * (unsigned)lw * 2u : 0x80000000u;
                        ncol = within;
                    }
                }
                if constexpr (MODE == OUT_QKV_IMAGE) {
                    // this wave's 32 columns of tile jt are exactly one head; the lane holds runs of 4 consecutive columns
                    // 8q + 4hf .. +3 = two rotation pairs with frequency indices 4q + 2hf and 4q + 2hf + 1
                    const int head = (n0 + 32 * jt) >> 5;
                    const bool rope = im.table && head < im.nq + im.nk;
                    const float sc = head < im.nq ? im.qscale : 1.0f;
                    const int mm = m < M ? m : M - 1;
                    // (a buffer resource + one 32-bit offset: the 64-bit row pointer cost the two registers this mode spilled, and a
                    // kernel that needs scratch pays ~5 us of idle queue on either side of every launch -- profiles/archive/r4_q_gaps.txt)
                    const int toff = (mm % im.S) * 128 + 16 * hf;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v0 = acc[jt][i][4 * q], v1 = acc[jt][i][4 * q + 1], v2 = acc[jt][i][4 * q + 2], v3 = acc[jt][i][4 * q + 3];
                        if (rope) {
                            const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(trs, toff + 32 * q, 0, 0));   // cos, sin, cos, sin
                            const float r0 = v0 * t.x - v1 * t.y, r1 = v1 * t.x + v0 * t.y;
                            const float r2 = v2 * t.z - v3 * t.w, r3 = v3 * t.z + v2 * t.w;
                            v0 = r0; v1 = r1; v2 = r2; v3 = r3;
                        }
                        acc[jt][i][4 * q] = v0 * sc; acc[jt][i][4 * q + 1] = v1 * sc;
                        acc[jt][i][4 * q + 2] = v2 * sc; acc[jt][i][4 * q + 3] = v3 * sc;
                    }
                }
                if constexpr (MODE == OUT_SWIGLU_BWD) {
                    // the lane's columns of this tile: 8q + 4hf .. + 3 (q = 0..3) of du, of a and of g; da | dg leave through the same
                    // half-wave pairing as every bf16 result.  Arithmetic of k_swiglu_bwd_bf16 (rowops.hip) on the ROUNDED du.
                    const unsigned grow = (m < M && wave_ok) ? (unsigned)m * (unsigned)im.F * 4u : 0x80000000u;
                    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                    auto rb = [](float v) { return __uint_as_float((unsigned)__builtin_bit_cast(bf16_t, (__bf16)v) << 16); };
                    const unsigned orow = (m < M && wave_ok) ? (unsigned)m * (unsigned)ldc * 2u : 0x80000000u;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {   // the runs (qq, qq + 2) leave together (half-wave pairing): 4 loads live at a time
                        u32x2v av[2], gv[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {   // one address register (row + half-wave), the column in the scalar offset
                            const int nsc = (ncol + 8 * (qq + 2 * h)) * 2;
                            av[h] = __builtin_bit_cast(u32x2v, __builtin_amdgcn_raw_buffer_load_b64(urs, grow + 8 * hf, nsc, 0));
                            gv[h] = __builtin_bit_cast(u32x2v, __builtin_amdgcn_raw_buffer_load_b64(urs, grow + 8 * hf, nsc + im.F * 2, 0));
                        }
                        unsigned pa[2][2], pg[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int q = qq + 2 * h;
                            float da[4], dg[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned aw = av[h][e >> 1], gw = gv[h][e >> 1];
                                const float a = __uint_as_float((e & 1) ? (aw & 0xffff0000u) : (aw << 16));
                                const float g = __uint_as_float((e & 1) ? (gw & 0xffff0000u) : (gw << 16));
                                const float d = rb(acc[jt][i][4 * q + e]);
                                const float sg = sigmoid_fast(a);
                                da[e] = d * g * sg * (1.f + a * (1.f - sg));
                                dg[e] = d * a * sg;
                            }
                            pa[h][0] = pack2(da[0], da[1]); pa[h][1] = pack2(da[2], da[3]);
                            pg[h][0] = pack2(dg[0], dg[1]); pg[h][1] = pack2(dg[2], dg[3]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pa[0][0], pa[1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pa[0][1], pa[1][1], false, false);
                        const u32x4 va = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pg[0][0], pg[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pg[0][1], pg[1][1], false, false);
                        const u32x4 vg = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                        const int nsc = (ncol + 8 * qq) * 2;
                        __builtin_amdgcn_raw_buffer_store_b128(va, crs, orow + 32 * hf, nsc, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(vg, crs, orow + 32 * hf, nsc + im.F * 2, 0);
                        __builtin_amdgcn_sched_barrier(0);   // keep the next runs' loads behind these stores: 256 registers, no scratch
                    }
                } else if constexpr (C16) {
                    unsigned pk[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pk[q][0] = pack2(acc[jt][i][4 * q], acc[jt][i][4 * q + 1]);
                        pk[q][1] = pack2(acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {   // runs (q, q+2): lower lanes end with columns 8q..8q+7, upper lanes with 16+8q..
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 2][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 2][1], false, false);
                        const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                        const int n = ncol + 8 * q + 16 * hf;
                        __builtin_amdgcn_raw_buffer_store_b128(v, crs, rowoff + n * 2, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = ncol + 8 * q + 4 * hf;
                        const f32x4 v = {acc[jt][i][4 * q], acc[jt][i][4 * q + 1], acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), crs, rowoff + n * 4, 0, 0);
                    }
                }
            }
            if constexpr (MODE == OUT_SWIGLU) {
                // u = silu(a) g from the ROUNDED a and g (what the stand-alone pass computes from the stored bf16 a | g)
                auto rb = [](float v) { return __uint_as_float((unsigned)__builtin_bit_cast(bf16_t, (__bf16)v) << 16); };
                unsigned pu[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = rb(acc[0][i][4 * q + e]), gv = rb(acc[1][i][4 * q + e]);
                        o[e] = av * sigmoid_fast(av) * gv;
                    }
                    pu[q][0] = pack2(o[0], o[1]);
                    pu[q][1] = pack2(o[2], o[3]);
                }
                const unsigned urow = (m < M && wave_ok) ? (unsigned)m * (unsigned)im.F * 2u : 0x80000000u;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pu[q][0], pu[q + 2][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pu[q][1], pu[q + 2][1], false, false);
                    const u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(v, urs, urow + (n0 + 8 * q + 16 * hf) * 2, 0, 0);
                }
            }
        }
    }
}

template <int MODE>
int launch_k256(const void* A, const void* W, void* C, int M, int N, int lda, int ldw, int ldc, const ImageArgs& im, hipStream_t st) {
    auto kern = k_gemm_k256<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) {
            gaot_set_error("gemm_k256: cannot set dynamic LDS %d: %s", 2 * STAGE, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int P = MODE == OUT_SWIGLU ? (im.F + 127) / 128 : (N + 255) / 256, nblk = (M + RB - 1) / RB;
    // ~512 workgroups (two per CU), but never more row chunks than row blocks
    int subs = std::max(1, 64 / P);
    subs = std::max(1, std::min(subs, (nblk + 7) / 8));
    GAOT_KLAUNCH(kern, dim3((unsigned)(8 * P * subs)), dim3(256), 2 * STAGE, st, (const bf16_t*)A, (const bf16_t*)W, C, M, N, lda, ldw, ldc,
                 P, subs, im);
    return GAOT_OK;
}

__global__ void k_rope_table(const float* __restrict__ freqs, int S, int half, float* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * half) return;
    const int pos = i / half, j = i % half;
    float sn, cs;
    sincosf((float)pos * freqs[j], &sn, &cs);   // the arithmetic of k_prep_qkv (attn_bf16.hip): same angles, same values
    table[2 * i] = cs;
    table[2 * i + 1] = sn;
}

}  // namespace

// 1 if the shape / layout is one this kernel takes (checked by gaot_gemm_bf16_dispatch before it falls to the generic tile kernel)
bool gaot_gemm_k256_applicable(const void* A, const void* W, const void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                               int64_t ldc, int c16) {
    const int64_t esz = c16 ? 2 : 4;
    return K == KK && M >= 1 && N >= 64 && N % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && (ldc * esz) % 16 == 0 &&
           ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0 && M * lda * 2 < 0x7fffffff &&
           M * ldc * esz < 0x7fffffff && lda >= KK && ldw >= KK;
}

int gaot_gemm_k256_launch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t lda, int64_t ldw, int64_t ldc,
                          int c16, hipStream_t st) {
    const ImageArgs none{nullptr, 1, 0, 0, 1.0f, 1, nullptr, 0};
    if (c16) return launch_k256<OUT_BF16>(A, W, C, (int)M, (int)N, (int)lda, (int)ldw, (int)ldc, none, st);
    return launch_k256<OUT_F32>(A, W, C, (int)M, (int)N, (int)lda, (int)ldw, (int)ldc, none, st);
}

// (cos, sin) of position * frequency for positions 0..S-1 and the 16 RoPE frequencies of a 32-wide head: [S][16][2] floats
extern "C" int gaot_rope_table(const float* freqs, int S, int half_dim, float* table, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(freqs && table && S > 0 && half_dim == 16, "bad argument (half_dim must be 16)");
    GAOT_KLAUNCH(k_rope_table, dim3((unsigned)ceil_div((int64_t)S * half_dim, 256)), dim3(256), 0, (hipStream_t)stream, freqs, S, half_dim, table);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// The fused q|k|v projection of one attention layer written straight as the attention kernels' bf16 image
// (reference attn.py:104-109: q_proj / k_proj / v_proj, rotary embedding of q and k; the 1/sqrt(d) log2(e) factor of the
// kernels folded into q): x [rows][256] bf16 (lda), w [(H + 2 HKV) * 32][256] bf16 (ldw), image [rows][(H + 2 HKV) * 32] bf16.
// rope_table: gaot_rope_table's output for this S, or NULL.  Replaces gaot_gemm_ex + the preparation pass of
// gaot_attn_fwd_bf16 (pass qkv = NULL there): the fp32 projection is never written.
extern "C" int gaot_qkv_image(const void* x_bf16, const void* w_bf16, void* image, int64_t rows, int64_t lda, int64_t ldw, int S,
                              int H, int HKV, const float* rope_table, float qscale, gaot_stream_t stream) {
    GAOT_ENTER();
    const int64_t N = (int64_t)(H + 2 * HKV) * 32;
    GAOT_CHECK_ARG(x_bf16 && w_bf16 && image && rows > 0 && S > 0 && H > 0 && HKV > 0, "bad argument");
    if (!gaot_gemm_k256_applicable(x_bf16, w_bf16, image, rows, N, KK, lda, ldw, N, 1)) {
        gaot_set_error("gaot_qkv_image: needs d_model = 256, 16-byte aligned bf16 rows and a heads * 32 that is a multiple of 64");
        return GAOT_ERR_UNSUPPORTED;
    }
    const ImageArgs im{rope_table, S, H, HKV, qscale, 1, nullptr, 0};
    return launch_k256<OUT_QKV_IMAGE>(x_bf16, w_bf16, image, (int)rows, (int)N, (int)lda, (int)ldw, (int)N, im, (hipStream_t)stream);
}

// The same projection for the sequence-parallel step (gaot_3d_amd/sharding.py): x holds THIS rank's token rows (global
// positions pos0 .. pos0 + rows - 1 of one sequence), and the image is written as `world` blocks [rows][(H + 2 HKV) / world * 32]
// -- block j = q | k | v of the heads rank j owns -- i.e. directly as the send buffer of the all-to-all that hands every rank
// all rows of its heads (what it receives IS its attention image).  rope_table: gaot_rope_table for the FULL sequence.
extern "C" int gaot_qkv_image_packed(const void* x_bf16, const void* w_bf16, void* packed, int64_t rows, int64_t lda, int64_t ldw,
                                     int64_t pos0, int H, int HKV, const float* rope_table, float qscale, int world,
                                     gaot_stream_t stream) {
    GAOT_ENTER();
    const int64_t N = (int64_t)(H + 2 * HKV) * 32;
    GAOT_CHECK_ARG(x_bf16 && w_bf16 && packed && rows > 0 && H > 0 && HKV > 0 && pos0 >= 0, "bad argument");
    GAOT_CHECK_ARG(world >= 1 && H % world == 0 && HKV % world == 0, "heads must divide over the ranks");
    if (!gaot_gemm_k256_applicable(x_bf16, w_bf16, packed, rows, N, KK, lda, ldw, N, 1)) {
        gaot_set_error("gaot_qkv_image_packed: needs d_model = 256, 16-byte aligned bf16 rows and a heads * 32 that is a multiple of 64");
        return GAOT_ERR_UNSUPPORTED;
    }
    // positions are pos0 + local row: the table pointer is advanced, S only has to exceed the local row count
    const ImageArgs im{rope_table ? rope_table + pos0 * 32 : nullptr, (int)rows, H, HKV, qscale, world, nullptr, 0};
    return launch_k256<OUT_QKV_IMAGE>(x_bf16, w_bf16, packed, (int)rows, (int)N, (int)lda, (int)ldw, (int)N, im, (hipStream_t)stream);
}

// The first half of the SwiGLU FFN in one launch (reference attn.py:155-156: silu(w1 x) * w3 x): x [rows][256] bf16, w13 = the
// co-located [w1; w3] ([2F][256] bf16); writes ag = w1 x | w3 x (bf16 [rows][2F], kept for the backward) and u = silu(a) g
// (bf16 [rows][F]) -- stands in for gaot_gemm_ex + gaot_swiglu_fwd_bf16 (one read of the 64 MB a | g tensor less per layer).
extern "C" int gaot_ffn_w13_swiglu(const void* x_bf16, const void* w13_bf16, void* ag, void* u, int64_t rows, int64_t lda, int64_t ldw,
                                   int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(x_bf16 && w13_bf16 && ag && u && rows > 0 && F > 0, "bad argument");
    if (F % 32 != 0 || !gaot_gemm_k256_applicable(x_bf16, w13_bf16, ag, rows, 2 * (int64_t)F, KK, lda, ldw, 2 * (int64_t)F, 1) ||
        ((uintptr_t)u % 16) != 0 || rows * (int64_t)F * 2 >= 0x7fffffff) {
        gaot_set_error("gaot_ffn_w13_swiglu: needs d_model = 256, F a multiple of 32 and 16-byte aligned bf16 buffers");
        return GAOT_ERR_UNSUPPORTED;
    }
    const ImageArgs im{nullptr, 1, 0, 0, 1.0f, 1, u, F};
    return launch_k256<OUT_SWIGLU>(x_bf16, w13_bf16, ag, (int)rows, 2 * F, (int)lda, (int)ldw, 2 * F, im, (hipStream_t)stream);
}

// The input gradient of the FFN's second projection with the SwiGLU backward in its epilogue (reference attn.py:155-157, autograd of
// w2(silu(w1 x) * w3 x)): dy [rows][256] bf16, w2t = W2^T ([F][256] bf16), ag = w1 x | w3 x (bf16 [rows][2F], saved by the forward)
// -> dag = d(a) | d(g) (bf16 [rows][2F]).  Stands in for gaot_gemm_ex (du = dy W2, bf16 result) + gaot_swiglu_bwd_bf16: du is
// rounded to bf16 exactly as the stored one was, so the values are the two-pass values.
extern "C" int gaot_ffn_w2_bwd_swiglu(const void* dy_bf16, const void* w2t_bf16, const void* ag, void* dag, int64_t rows, int64_t lda,
                                      int64_t ldw, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(dy_bf16 && w2t_bf16 && ag && dag && rows > 0 && F > 0, "bad argument");
    if (F % 64 != 0 || !gaot_gemm_k256_applicable(dy_bf16, w2t_bf16, dag, rows, (int64_t)F, KK, lda, ldw, 2 * (int64_t)F, 1) ||
        ((uintptr_t)ag % 16) != 0 || rows * (int64_t)F * 4 >= 0x7fffffff) {
        gaot_set_error("gaot_ffn_w2_bwd_swiglu: needs d_model = 256, F a multiple of 64 and 16-byte aligned bf16 buffers");
        return GAOT_ERR_UNSUPPORTED;
    }
    const ImageArgs im{nullptr, 1, 0, 0, 1.0f, 1, const_cast<void*>(ag), F};
    return launch_k256<OUT_SWIGLU_BWD>(dy_bf16, w2t_bf16, dag, (int)rows, F, (int)lda, (int)ldw, 2 * F, im, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// C[M,256] = A[M,K] . W[256,K]^T (+ residual), fp32 result, both operands bf16 in memory and k-contiguous, K a multiple of 64:
// the d_model-wide outputs of the FFN (w2 forward, K = F; input gradient of w1|w3 on the transposed weight, K = 2F).  Same
// skeleton as k_gemm_k256 -- LDS-DMA, source-side swizzle, one barrier per step, transposed product so that a lane owns a
// row and stores 16 bytes -- but K is long, so the weight tile streams too: per 64-deep step a workgroup (64 rows x all 256
// columns) DMAs 8 KB of A and 32 KB of W into one of two LDS buffers while the MFMAs of the previous step run.
// ------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int TB_W = 256 * 128;   // weight tile of one step: 256 rows x 64 k bf16 = 32 KB

// MT = 32-row tiles per workgroup along M (2: 64 rows; 4: 128 rows -- half as many weight bytes per output row but half as many
// workgroups: measured slower, the kernel is bound by the DMA latency of each CU, not by the aggregate L2 traffic)
template <bool HAS_RES, int MT, int NB>
__global__ __launch_bounds__(256, 1) void k_gemm_tn_n256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                        float* __restrict__ C, const float* __restrict__ R, int M, int K,
                                                                        int lda, int ldw, int ldc, int ldr) {
    constexpr int BM = 32 * MT, TB_A = BM * 128, TB_STAGE = TB_A + TB_W;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    // consecutive workgroup ids go to consecutive XCDs: give each XCD a contiguous range of row blocks
    const int nblk = (M + BM - 1) / BM;
    const int per = (nblk + 7) / 8;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= nblk) return;
    const int m0 = t * BM;
    const int64_t abytes = (int64_t)M * lda * 2, wbytes = (int64_t)256 * ldw * 2, cbytes = (int64_t)M * ldc * 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(abytes > 0x7fffffff ? 0x7fffffff : abytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(wbytes > 0x7fffffff ? 0x7fffffff : wbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(cbytes > 0x7fffffff ? 0x7fffffff : cbytes), 0x00020000);
    // one DMA piece = 1 KB = 8 tile rows of 128 B (64 k); lane -> (row = lane >> 3, slot = lane & 7); slot s of row r holds
    // source chunk s ^ (r & 7), so a fragment read of chunk c goes to slot c ^ (r & 7): conflict-free ds_read_b128
    const int prow = lane >> 3, pslot = lane & 7;
    auto stage = [&](int k0, int buf) {
        char* base = lds + buf * TB_STAGE;
        // A: BM / 8 pieces, wave w takes MT of them; W: 32 pieces (256 rows), wave w takes 8w .. 8w+7
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int piece = wave * MT + i, row = piece * 8 + prow;
            const int voff = (m0 + row) * lda * 2 + ((pslot ^ (row & 7)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, voff, k0 * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = wave * 8 + i, row = piece * 8 + prow;
            const int voff = row * ldw * 2 + ((pslot ^ (row & 7)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(base + TB_A + piece * 1024), 16, voff, k0 * 2, 0, 0);
        }
    };
    f32x16 acc[2][MT];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
    // the residual rows of this lane's outputs are requested HERE, before the first DMA piece: they are the oldest entries of
    // the in-order vmcnt queue (the counted waits of the K loop are unaffected) and have landed long before the epilogue.  In the
    // epilogue (`if (ok) v += *R`) every one of the 8 MT loads sat in a block of its own with an s_waitcnt vmcnt(0) behind it:
    // 8 MT serialised L2 round trips per workgroup.
    f32x4 rres[HAS_RES ? MT : 1][2][4];
    if constexpr (HAS_RES) {
        const int64_t rbytes = (int64_t)M * ldr * 4;
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)(rbytes > 0x7fffffff ? 0x7fffffff : rbytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + 32 * i + l31;
            const unsigned roff = m < M ? (unsigned)m * (unsigned)ldr * 4u : 0x80000000u;     // past M: out of range -> zeros
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    rres[i][jt][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + (wave * 64 + 32 * jt + 8 * q + 4 * hf) * 4, 0, 0));
        }
    }
    const int nsteps = K / 64;
    // NB LDS buffers: NB - 1 steps of DMA in flight (each wave issues MT + 8 pieces per step; the kernel is bound by the latency
    // of these loads -- ~12 B/clk per CU with one step in flight -- so depth pays: 2 -> 3 buffers -0.18 ms per step)
    constexpr int PIECES = MT + 8;
#pragma unroll
    for (int p = 0; p < NB - 1; ++p)
        if (p < nsteps) stage(p * 64, p);
    if (nsteps >= NB - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int sw = l31 & 7;
    for (int kt = 0; kt < nsteps; ++kt) {
        const int buf = kt % NB;
        __builtin_amdgcn_s_barrier();
        if (kt + NB - 1 < nsteps) stage((kt + NB - 1) * 64, (kt + NB - 1) % NB);
        const char* ab = lds + buf * TB_STAGE + l31 * 128;                       // activation rows 0..31 (tile i: + 4096 i)
        const char* wb = lds + buf * TB_STAGE + TB_A + (wave * 64 + l31) * 128;    // this wave's 64 weight rows (+32 rows: + 4096)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = ((2 * s + hf) ^ sw) << 4;
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wb + slot), w1 = *reinterpret_cast<const bf16x8*>(wb + 4096 + slot);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ab + 4096 * i + slot);
                acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a, acc[0][i], 0, 0, 0);
                acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a, acc[1][i], 0, 0, 0);
            }
        }
        // step kt + 1 must have landed before the next barrier; the NB - 2 steps issued after it may stay in flight
        if (kt + NB - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // acc[jt][i][r]: column n = 64 wave + 32 jt + mfma32_row(r, hf), row m = m0 + 32 i + l31
    const int n0 = wave * 64;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + 32 * i + l31;
        const bool ok = m < M;
        const unsigned rowoff = ok ? (unsigned)m * (unsigned)ldc * 4u : 0x80000000u;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 32 * jt + 8 * q + 4 * hf;
                f32x4 v = {acc[jt][i][4 * q], acc[jt][i][4 * q + 1], acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]};
                if constexpr (HAS_RES) v += rres[i][jt][q];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), crs, rowoff + n * 4, 0, 0);
            }
    }
}

template <bool HAS_RES, int MT, int NB>
int launch_tn_n256(const void* A, const void* W, float* C, const float* R, int M, int K, int lda, int ldw, int ldc, int ldr, hipStream_t st) {
    constexpr int BM = 32 * MT, LDS = NB * (BM * 128 + TB_W);
    auto kern = k_gemm_tn_n256<HAS_RES, MT, NB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            gaot_set_error("gemm_tn_n256: cannot set dynamic LDS %d: %s", LDS, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int nblk = (M + BM - 1) / BM, per = (nblk + 7) / 8;
    GAOT_KLAUNCH(kern, dim3((unsigned)(8 * per)), dim3(256), LDS, st, (const bf16_t*)A, (const bf16_t*)W, C, R, M, K, lda, ldw, ldc, ldr);
    return GAOT_OK;
}

}  // namespace

bool gaot_gemm_tn_n256_applicable(const void* A, const void* W, const void* C, const void* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                                  int64_t ldw, int64_t ldc, int64_t ldr) {
    return N == 256 && K >= 512 && K % 64 == 0 && M >= 1 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && (!R || ldr % 4 == 0) &&
           ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)R % 16) == 0 &&
           M * lda * 2 < 0x7fffffff && 256 * ldw * 2 < 0x7fffffff && M * ldc * 4 < 0x7fffffff && (!R || M * ldr * 4 < 0x7fffffff) &&
           lda >= K && ldw >= K;   // every operand is addressed through a buffer resource with 32-bit byte offsets
}

int gaot_gemm_tn_n256_launch(const void* A, const void* W, float* C, const float* R, int64_t M, int64_t K, int64_t lda, int64_t ldw,
                             int64_t ldc, int64_t ldr, hipStream_t st) {
    // ring of three LDS buffers (two steps of DMA in flight): 2 -> 3 buffers -0.18 ms per step, 4 nothing more (round 2)
    if (R) return launch_tn_n256<true, 2, 3>(A, W, C, R, (int)M, (int)K, (int)lda, (int)ldw, (int)ldc, (int)ldr, st);
    return launch_tn_n256<false, 2, 3>(A, W, C, R, (int)M, (int)K, (int)lda, (int)ldw, (int)ldc, 0, st);
}
