// LAB (round 6; measured, not shipped -- DESIGN.md §8.3, profiles/r6_zg ... r6_zj; built by tools/lab/dw_frag_lab.py).
// Weight-gradient products dW[n1][n2] = sum_m A[m][n1] B[m][n2] of the latent Transformer (reference attn.py:110-157: the autograd of
// q | k | v, o_proj, w1 | w3, w2 -- nn.Linear.weight.grad) over operands that arrive ALREADY IN MFMA-FRAGMENT ORDER.
//
// Both operands of a weight gradient are reduction-strided in memory (the reduction runs over the ROWS m of two row-major activation
// matrices).  The generic kernel (gemm_bf16.hip, A_KS && B_KS) stages [64 m][128 n] tiles in LDS and reads them back with
// ds_read_b64_tr_b16, between two barriers per 64 rows: 400 TF/s, 3-4x its HBM time.  Here the PRODUCERS of the operands (the row-block
// kernels of ffn_fused.hip, which own 64 whole rows at a time and hold them in LDS anyway) write a second form of them, the
//     T-image of X [M][N] (bf16):  1-KB blocks;  block (g, ks, c): columns 128 g + 32 c .. + 31, rows 16 ks .. + 15;
//                                  lane (l31, hf) holds 8 bf16: X[16 ks + 8 (j >> 2) + 4 hf + (j & 3)][128 g + 32 c + l31], j = 0..7
//                                  (the row order ds_read_b64_tr_b16 delivers; the same for both operands, so the product does not care)
//                                  at byte ((g * KS + ks) * 4 + c) * 1024 + 16 lane,   KS = rows / 16 (rows padded to 64 with zeros)
// i.e. exactly what v_mfma_f32_32x32x16_bf16 takes as an operand.  A column group's k-steps are contiguous (4 KB each): a wave streams
// its operands global -> VGPR like a copy, no LDS, no barrier, no transposed read in this kernel.
// Work split: a workgroup owns one 128 x 128 output tile over a k-range; its four waves take a quarter of the range each with the WHOLE
// tile in their accumulators (256 registers; 8 KB of operands per 16 MFMAs), their four tiles are summed through LDS in a fixed order
// and leave as ONE fp32 partial [split][N1][N2] -- gaot_reduce_multi (or the in-call pass) sums the splits.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(bf16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)b) << 16);
}

// row-major -> T-image (tests, labs and operands no row-block kernel produces): one thread per (block, lane)
template <typename T>
__global__ void k_timg_pack(const T* __restrict__ X, int64_t ld, int M, int N, u32x4* __restrict__ out) {
    const int KS = ((M + 63) / 64) * 4;
    const int64_t total = (int64_t)(N / 128) * KS * 4 * 64;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(id & 63), c = (int)((id >> 6) & 3);
        const int64_t gk = id >> 8;
        const int ks = (int)(gk % KS), g = (int)(gk / KS);
        const int col = 128 * g + 32 * c + (lane & 31), hf = lane >> 5;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
            v[j] = m < M ? (float)X[(int64_t)m * ld + col] : 0.f;
        }
        out[id] = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    }
}

struct DwArgs {
    const u32x4* A; const u32x4* B; float* part;
    int KS, n1g, n2g, splits;
    int abl;      // lab only (GAOT_DW_FRAG_ABL): 1 = no cross-wave sum (wrong results: timing of the tail), 2 = no operand loads after the first
};

constexpr int DW_LDS = 2 * 65536;

// RD - 1 = k-steps of operands in flight per wave (8 KB each)
template <int RD>
__global__ __launch_bounds__(256, 1) void k_dw_frag(DwArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // consecutive workgroup ids sit on consecutive XCDs: with splits a multiple of 8 a split (= a k-range of BOTH operands) stays on
    // one XCD, whose L2 then serves the B slab to every output tile of that range
    const int split = blockIdx.x % a.splits, tile = blockIdx.x / a.splits;
    const int i1 = tile / a.n2g, i2 = tile % a.n2g;
    const int KS = a.KS;
    const int per = (((KS + a.splits - 1) / a.splits + 3) / 4) * 4, q4 = per / 4;
    const int k0 = split * per + wv * q4;
    const int k1 = min(min(KS, (split + 1) * per), k0 + q4);
    const unsigned nrec = (unsigned)KS * 4096u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)i1 * KS * 256), 0, (int)nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)i2 * KS * 256), 0, (int)nrec, 0x00020000);
    u32x4 fa[RD][4], fb[RD][4];
    auto load = [&](int slot, int ks) {
        const int so = __builtin_amdgcn_readfirstlane(ks < k1 ? ks * 4096 : (int)nrec);     // past the wave's range: out of the buffer -> zeros
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fa[slot][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, so + c * 1024, 0));
            fb[slot][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, so + c * 1024, 0));
        }
    };
    f32x16 acc[4][4];       // [j: 32 columns of dW][i: 32 rows of dW]; lane = row 32 i + l31, register r = column 32 j + 8 (r >> 2) + 4 hf + (r & 3)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
#pragma unroll
    for (int s = 0; s < RD - 1; ++s) {
        load(s, k0 + s);
        __builtin_amdgcn_sched_barrier(0);      // in this order: the counted waits below rely on it
    }
    for (int ks = k0; ks < k1; ks += RD) {
#pragma unroll
        for (int s = 0; s < RD; ++s) {
            if (!(a.abl & 2)) load((s + RD - 1) % RD, ks + s + RD - 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (RD - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[s][j]), __builtin_bit_cast(bf16x8, fa[s][i]), acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the four waves' tiles -> one, ((t0 + t1) + t2) + t3.  No wave ever adds INTO its accumulators (they would have to leave the
    // accumulation registers for that): two at a time the waves write their tiles into two 64-KB regions (lane-linear 16-byte slots:
    // (j, i, q) at ((4 j + i) * 4 + q) * 1024 + 16 lane), and wave w sums columns 32 w .. + 31 (j = w) of what it reads back.
    auto put = [&](int region) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(lds + region * 65536 + (((4 * j + i) * 4 + q) << 10) + lane * 16) =
                        f32x4{acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]};
    };
    const char* mine = lds + (wv << 14) + lane * 16;      // quarter j = wv of region 0
    f32x4 sum[16];
    if (a.abl & 1) {
        if (wv == 0) put(0);
        __syncthreads();
    } else {
    if (wv < 2) put(wv);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) sum[t] = *reinterpret_cast<const f32x4*>(mine + (t << 10)) + *reinterpret_cast<const f32x4*>(mine + 65536 + (t << 10));
    __syncthreads();
    if (wv >= 2) put(wv - 2);
    __syncthreads();
    }
    const int N2 = a.n2g * 128;
    float* out = a.part + (int64_t)split * (a.n1g * 128) * N2 + (int64_t)(i1 * 128) * N2 + i2 * 128;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 128 * N2 * 4, 0x00020000);
    // one address register: lane (l31, hf) -> row l31, 16 bytes at column 4 hf; row block i, quarter wv and q are scalar offsets
    const int voff = (l31 * N2 + 4 * hf) * 4;
#pragma unroll
    for (int t = 0; t < 16; ++t) {      // t = 4 i + q
        const f32x4 o = (sum[t] + *reinterpret_cast<const f32x4*>(mine + (t << 10))) + *reinterpret_cast<const f32x4*>(mine + 65536 + (t << 10));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ors, voff, (32 * (t >> 2) * N2 + 32 * wv + 8 * (t & 3)) * 4, 0);
    }
}
}  // namespace

// bytes of the T-image of an operand with `rows` rows and `cols` columns (cols a multiple of 128)
extern "C" int64_t gaot_timg_bytes(int64_t rows, int64_t cols) { return ((rows + 63) / 64) * 64 * cols * 2; }

// row-major fp32 / bf16 [rows][cols] (leading dimension ld, in elements) -> T-image (see the head of this file)
extern "C" int gaot_timg_pack(const void* x, int is_bf16, int64_t ld, int64_t rows, int64_t cols, void* image, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(x && image && rows > 0 && cols > 0 && cols % 128 == 0 && ld >= cols && ((uintptr_t)image % 16) == 0, "bad argument (cols must be a multiple of 128)");
    GAOT_CHECK_ARG(rows < (1 << 30), "too many rows");
    const int64_t total = gaot_timg_bytes(rows, cols) / 16;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    if (is_bf16) {
        GAOT_KLAUNCH((k_timg_pack<__bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ld, (int)rows, (int)cols, (u32x4*)image);
    } else {
        GAOT_KLAUNCH((k_timg_pack<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, (int)rows, (int)cols, (u32x4*)image);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// number of partial products gaot_dw_frag writes for this shape (each [n1][n2] fp32); a multiple of 8
extern "C" int gaot_dw_frag_splits(int64_t rows, int64_t n1, int64_t n2) {
    if (rows <= 0 || n1 <= 0 || n2 <= 0 || n1 % 128 || n2 % 128) return 0;
    const int64_t tiles = (n1 / 128) * (n2 / 128), ks = ((rows + 63) / 64) * 4;
    static int cap = -1;      // lab switch GAOT_DW_FRAG_MAXSPLITS
    if (cap < 0) {
        const char* e = getenv("GAOT_DW_FRAG_MAXSPLITS");
        cap = e ? atoi(e) : 0;
    }
    int64_t s = std::max<int64_t>(8, (256 / tiles) / 8 * 8);
    if (cap >= 8) s = std::min<int64_t>(s, cap / 8 * 8);
    while (s > 8 && s * 4 > ks) s -= 8;      // at least one k-step per wave
    return (int)s;
}

// dW partials from two T-images over the same rows: part[s][n1][n2] (fp32, s < gaot_dw_frag_splits(rows, n1, n2)), whose fixed-order
// sum over s is dW = A^T B.  a_image / b_image: T-images of A [rows][n1] and B [rows][n2].
extern "C" int gaot_dw_frag(const void* a_image, const void* b_image, int64_t rows, int64_t n1, int64_t n2, float* part, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(a_image && b_image && part && rows > 0 && n1 > 0 && n2 > 0 && n1 % 128 == 0 && n2 % 128 == 0, "bad argument (n1, n2 must be multiples of 128)");
    GAOT_CHECK_ARG(((uintptr_t)a_image % 16) == 0 && ((uintptr_t)b_image % 16) == 0 && ((uintptr_t)part % 16) == 0, "16-byte alignment");
    const int64_t ks = ((rows + 63) / 64) * 4;
    if (ks * 4096 >= 0x7fffffff) {
        gaot_set_error("gaot_dw_frag: too many rows for 32-bit buffer offsets");
        return GAOT_ERR_UNSUPPORTED;
    }
    static int rd = 0;      // lab switch GAOT_DW_RD (3..7), read once
    if (!rd) {
        const char* e = getenv("GAOT_DW_RD");
        rd = e ? atoi(e) : 5;
        if (rd < 3 || rd > 7) rd = 5;
        hipError_t err = hipSuccess;
        if (rd == 3) err = hipFuncSetAttribute((const void*)k_dw_frag<3>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (rd == 4) err = hipFuncSetAttribute((const void*)k_dw_frag<4>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (rd == 5) err = hipFuncSetAttribute((const void*)k_dw_frag<5>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (rd == 6) err = hipFuncSetAttribute((const void*)k_dw_frag<6>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (rd == 7) err = hipFuncSetAttribute((const void*)k_dw_frag<7>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (err != hipSuccess) {
            rd = 0;
            gaot_set_error("dw_frag: cannot set dynamic LDS %d: %s", DW_LDS, hipGetErrorString(err));
            return GAOT_ERR_LAUNCH;
        }
    }
    const int splits = gaot_dw_frag_splits(rows, n1, n2);
    static int abl = -1;
    if (abl < 0) {
        const char* e = getenv("GAOT_DW_FRAG_ABL");
        abl = e ? atoi(e) : 0;
    }
    const DwArgs a{(const u32x4*)a_image, (const u32x4*)b_image, part, (int)ks, (int)(n1 / 128), (int)(n2 / 128), splits, abl};
    const dim3 grid((unsigned)(a.n1g * a.n2g * splits));
    switch (rd) {
        case 3: GAOT_KLAUNCH(k_dw_frag<3>, grid, dim3(256), DW_LDS, (hipStream_t)stream, a); break;
        case 4: GAOT_KLAUNCH(k_dw_frag<4>, grid, dim3(256), DW_LDS, (hipStream_t)stream, a); break;
        case 5: GAOT_KLAUNCH(k_dw_frag<5>, grid, dim3(256), DW_LDS, (hipStream_t)stream, a); break;
        case 6: GAOT_KLAUNCH(k_dw_frag<6>, grid, dim3(256), DW_LDS, (hipStream_t)stream, a); break;
        default: GAOT_KLAUNCH(k_dw_frag<7>, grid, dim3(256), DW_LDS, (hipStream_t)stream, a); break;
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
