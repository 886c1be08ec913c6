// bf16 matrix-core GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 I/O) for the Transformer
// projections and their autograd -- precision 1 of gaot_gemm (see gemm.hip for the operator contract).
//
//   C[m][n] = act( sum_k A(m,k) B(k,n) + bias[n] ) + residual[m][n]
//
// fp32 operands are rounded to bf16 once, while they are staged into LDS.  An operand whose reduction
// index is contiguous in memory (x in x W^T, W in x W^T, dy in dy W) is staged as [row][64 k] and read
// with ds_read_b128; an operand whose reduction index is the SLOW index (W in dy W, both operands of
// the weight gradient dy^T x) is staged in its memory orientation [64 k][row] and read through the
// hardware-transposing ds_read_b64_tr_b16 -- no scattered 2-byte LDS writes, no transposed copies in
// HBM.  Row pitches are padded (144 B / 320 B) so that both kinds of read are bank-conflict free.
// 128x128 tile, 4 waves (2x2), 64x64 per wave, BK = 64, register-staged prefetch of the next tile.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BN = 128, BK = 64;   // BM is a template parameter: 128, or 64 when the grid would under-fill the chip
typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

struct GArgs {
    const void* A; const void* B; void* C; float* preact; const float* bias; const float* residual;
    int64_t M, N, K, lda, ldb, ldc, ldr;
    int act, splits;
    int64_t k_per_split;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(bf16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)b) << 16);
}
__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 1: return gelu_f(v);
        case 2: return v > 0.f ? v : 0.f;
        case 3: return v / (1.f + __expf(-v));
        default: return v;
    }
}

// KS == false: operand element (row, k) at X[row*ld + k]  -> LDS [128 rows][64 k], pitch 144 B
// KS == true : operand element (row, k) at X[k*ld + row]  -> LDS [64 k][128 rows], pitch 320 B
template <bool KS, int ROWS>
struct Tile {
    static constexpr int PITCH = KS ? (ROWS * 2 + 64) : (BK * 2 + 16);
    static constexpr int BYTES = KS ? BK * PITCH : ROWS * PITCH;
    static constexpr int NV = ROWS / 16;   // float4 per thread per tile
    float4 regs[NV];

    __device__ __forceinline__ void load(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                         int64_t kend, bool vec_ok) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = threadIdx.x + v * 256;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!KS) {
                const int r = idx >> 4, kq = idx & 15;   // 16 float4 per row of 64 k
                const int64_t row = row0 + r, k = k0 + 4 * kq;
                if (row < nrows && k < kend) {
                    const float* p = X + row * ld + k;
                    if (vec_ok && k + 3 < kend) t = *reinterpret_cast<const float4*>(p);
                    else {
                        t.x = p[0];
                        if (k + 1 < kend) t.y = p[1];
                        if (k + 2 < kend) t.z = p[2];
                        if (k + 3 < kend) t.w = p[3];
                    }
                }
            } else {
                const int kk = idx / (ROWS / 4), rq = idx % (ROWS / 4);  // ROWS/4 float4 per k-row
                const int64_t k = k0 + kk, row = row0 + 4 * rq;
                if (k < kend && row < nrows) {
                    const float* p = X + k * ld + row;
                    if (vec_ok && row + 3 < nrows) t = *reinterpret_cast<const float4*>(p);
                    else {
                        t.x = p[0];
                        if (row + 1 < nrows) t.y = p[1];
                        if (row + 2 < nrows) t.z = p[2];
                        if (row + 3 < nrows) t.w = p[3];
                    }
                }
            }
            regs[v] = t;
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = threadIdx.x + v * 256;
            const uint2 pk = make_uint2(pack2(regs[v].x, regs[v].y), pack2(regs[v].z, regs[v].w));
            if (!KS) {
                const int r = idx >> 4, kq = idx & 15;
                *reinterpret_cast<uint2*>(lds + r * PITCH + kq * 8) = pk;
            } else {
                const int kk = idx / (ROWS / 4), rq = idx % (ROWS / 4);
                *reinterpret_cast<uint2*>(lds + kk * PITCH + rq * 8) = pk;
            }
        }
    }
    // MFMA operand fragment for tile-row block `rb` (32 rows), k-step s (16 k): element j <-> k index
    //   KS == false: k = 16s + 8hf + j                          (ds_read_b128 along k)
    //   KS == true : k = 16s + 8(j>>2) + 4hf + (j&3)            (two transposed 4x16 block reads)
    // Both operands of one GEMM may use different k orders only if ... they may not: see k_gemm_bf16.
    static __device__ __forceinline__ bf16x8 frag(const char* lds, int rb, int s, int lane) {
        const int l31 = lane & 31, hf = lane >> 5;
        if (!KS) {
            return *reinterpret_cast<const bf16x8*>(lds + (rb * 32 + l31) * PITCH + (16 * s + 8 * hf) * 2);
        } else {
            const int i = lane & 15, grp = (lane >> 4) & 1;
            const int col = rb * 32 + 16 * grp + 4 * (i & 3);
            const int r0 = 16 * s + 4 * hf + (i >> 2);
            const char* p0 = lds + r0 * PITCH + col * 2;
            const char* p1 = p0 + 8 * PITCH;
            const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
            const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
            bf16x8 o;
            o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
            o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
            return o;
        }
    }
    // same k order as the KS fragment, but from a k-contiguous tile: gather k = 16s+4hf+{0..3} and +8
    static __device__ __forceinline__ bf16x8 frag_as_ks(const char* lds, int rb, int s, int lane) {
        const int l31 = lane & 31, hf = lane >> 5;
        const char* p = lds + (rb * 32 + l31) * PITCH + (16 * s + 4 * hf) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(p);
        const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
        bf16x8 o;
        o[0] = (short)(lo.x & 0xffff); o[1] = (short)(lo.x >> 16); o[2] = (short)(lo.y & 0xffff); o[3] = (short)(lo.y >> 16);
        o[4] = (short)(hi.x & 0xffff); o[5] = (short)(hi.x >> 16); o[6] = (short)(hi.y & 0xffff); o[7] = (short)(hi.y >> 16);
        return o;
    }
};

// Branch-free variant of Tile::load for 16-byte aligned operands with K % 4 == 0: the tile is addressed through
// a buffer resource that starts at the tile's first row, so rows past the end of the matrix read as zero in
// hardware and the k tail is masked by pushing the lane's offset out of range.  One VGPR of address per operand;
// the per-load row / k displacement is a scalar offset.
template <bool KS, int ROWS>
struct VTile : Tile<KS, ROWS> {
    using Base = Tile<KS, ROWS>;
    static constexpr int NV = Base::NV;
    static constexpr int KR = 1024 / ROWS;   // KS: k-rows covered by one pass of the 256 threads
    __amdgpu_buffer_rsrc_t rs;
    int voff, ld4, kk;

    __device__ __forceinline__ void setup(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t K) {
        const float* base = KS ? X + row0 : X + row0 * ld;
        int64_t bytes = KS ? ((K - 1) * ld + (nrows - row0)) * 4 : ((nrows - row0 - 1) * ld + K) * 4;
        bytes = bytes < 0 ? 0 : (bytes > 0x7fffffff ? 0x7fffffff : bytes);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
        ld4 = (int)ld * 4;
        if (!KS) {
            kk = 4 * (threadIdx.x & 15);
            voff = (threadIdx.x >> 4) * ld4 + kk * 4;
        } else {
            kk = threadIdx.x / (ROWS / 4);
            voff = kk * ld4 + (threadIdx.x % (ROWS / 4)) * 16;
        }
    }
    __device__ __forceinline__ void load(int k0, int kend) {
        if (!KS) {
            const int vo = (k0 + kk < kend) ? voff : (int)0x80000000;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                this->regs[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, v * 16 * ld4 + k0 * 4, 0));
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int vo = (k0 + v * KR + kk < kend) ? voff : (int)0x80000000;
                this->regs[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (k0 + v * KR) * ld4, 0));
            }
        }
    }
};

// The same tile for an operand that already IS bf16 in memory (an activation written by a previous kernel of the
// bf16 path): half the bytes per element on the way in, no conversion, 16-byte LDS stores.  8 elements per 16-B
// chunk: rows / k-rows must be multiples of 8 elements and 16-byte aligned.
template <bool KS, int ROWS>
struct VTile16 : Tile<KS, ROWS> {
    using Base = Tile<KS, ROWS>;
    static constexpr int NV = ROWS / 32;       // 16-B chunks per thread per tile
    static constexpr int CPR = ROWS / 8;       // KS: chunks per k-row
    static constexpr int KR = 256 / CPR;       // KS: k-rows covered by one pass of the 256 threads
    __amdgpu_buffer_rsrc_t rs;
    int voff, ld2, kk;
    uint4 r16[NV];

    __device__ __forceinline__ void setup(const void* __restrict__ Xv, int64_t ld, int64_t row0, int64_t nrows, int64_t K) {
        const unsigned short* X = (const unsigned short*)Xv;
        const unsigned short* base = KS ? X + row0 : X + row0 * ld;
        int64_t bytes = KS ? ((K - 1) * ld + (nrows - row0)) * 2 : ((nrows - row0 - 1) * ld + K) * 2;
        bytes = bytes < 0 ? 0 : (bytes > 0x7fffffff ? 0x7fffffff : bytes);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
        ld2 = (int)ld * 2;
        if (!KS) {
            kk = 8 * (threadIdx.x & 7);
            voff = (threadIdx.x >> 3) * ld2 + kk * 2;
        } else {
            kk = threadIdx.x / CPR;
            voff = kk * ld2 + (threadIdx.x % CPR) * 16;
        }
    }
    __device__ __forceinline__ void load(int k0, int kend) {
        if (!KS) {
            const int vo = (k0 + kk < kend) ? voff : (int)0x80000000;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                r16[v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, v * 32 * ld2 + k0 * 2, 0));
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int vo = (k0 + v * KR + kk < kend) ? voff : (int)0x80000000;
                r16[v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (k0 + v * KR) * ld2, 0));
            }
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!KS) *reinterpret_cast<uint4*>(lds + ((threadIdx.x >> 3) + v * 32) * Base::PITCH + (threadIdx.x & 7) * 16) = r16[v];
            else *reinterpret_cast<uint4*>(lds + (kk + v * KR) * Base::PITCH + (threadIdx.x % CPR) * 16) = r16[v];
        }
    }
};

template <bool A_KS, bool B_KS, int BM, bool VEC, int OCC, bool A16 = false, bool B16 = false, bool C16 = false>
__global__ __launch_bounds__(256, OCC) void k_gemm_bf16(GArgs g, int a_vec, int b_vec) {
    static_assert(VEC || !(A16 || B16), "bf16 operands in memory only on the buffer-addressed path");
    using TA = typename std::conditional<A16, VTile16<A_KS, BM>, typename std::conditional<VEC, VTile<A_KS, BM>, Tile<A_KS, BM>>::type>::type;
    using TB = typename std::conditional<B16, VTile16<B_KS, BN>, typename std::conditional<VEC, VTile<B_KS, BN>, Tile<B_KS, BN>>::type>::type;
    constexpr int MT = BM / 64;   // 32-row MFMA tiles per wave along M (waves are 2 x 2)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* la = lds;
    char* lb = lds + TA::BYTES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // row-block-fastest workgroup order: consecutive ids (-> consecutive XCDs) take consecutive row blocks of the SAME
    // column block, so an XCD keeps re-using its own 1/8 of A (2 MB at M = 16 384, K = 256) from its L2 across all
    // column blocks, and the column block's weight slice is shared by everyone
    // The split-K weight gradients (A_KS) are launched as a 1-D grid: all row blocks of one (split, column block) group
    // read the same k-slab of B, so a whole group is given to ONE XCD (id % 8) when the group count allows it.
    int bxm, bxn, bz;
    if constexpr (A_KS) {
        const int nbm = (int)((g.M + BM - 1) / BM), nbn = (int)((g.N + BN - 1) / BN);
        const int i = blockIdx.x, groups = nbn * g.splits;
        if (groups % 8 == 0) {
            const int xcd = i & 7, j = i >> 3;
            const int grp = (j / nbm) * 8 + xcd;
            bxm = j % nbm; bxn = grp % nbn; bz = grp / nbn;
        } else {
            bxn = i % nbn; bxm = (i / nbn) % nbm; bz = i / (nbn * nbm);
        }
    } else {
        bxm = blockIdx.x; bxn = blockIdx.y; bz = blockIdx.z;
    }
    const int64_t bm = (int64_t)bxm * BM, bn = (int64_t)bxn * BN;
    const int64_t kbeg = (int64_t)bz * g.k_per_split;
    const int64_t kend = (g.splits > 1) ? ((kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K) : g.K;
    // the two operands must walk k in the same order inside a 16-step: if exactly one of them is k-strided
    // (transposed reads), the k-contiguous one gathers its 8 elements in that order as well
    constexpr bool MIXED = A_KS != B_KS;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TA ta;
    TB tb;
    if constexpr (VEC) {
        if constexpr (A16) ta.setup(g.A, g.lda, bm, g.M, g.K);
        else ta.setup((const float*)g.A, g.lda, bm, g.M, g.K);
        if constexpr (B16) tb.setup(g.B, g.ldb, bn, g.N, g.K);
        else tb.setup((const float*)g.B, g.ldb, bn, g.N, g.K);
        if (kbeg < kend) {
            ta.load((int)kbeg, (int)kend);
            tb.load((int)kbeg, (int)kend);
        }
    } else if (kbeg < kend) {
        ta.load((const float*)g.A, g.lda, bm, g.M, kbeg, kend, a_vec);
        tb.load((const float*)g.B, g.ldb, bn, g.N, kbeg, kend, b_vec);
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        ta.store(la);
        tb.store(lb);
        __syncthreads();
        if (k0 + BK < kend) {
            if constexpr (VEC) {
                ta.load((int)k0 + BK, (int)kend);
                tb.load((int)k0 + BK, (int)kend);
            } else {
                ta.load((const float*)g.A, g.lda, bm, g.M, k0 + BK, kend, a_vec);
                tb.load((const float*)g.B, g.ldb, bn, g.N, k0 + BK, kend, b_vec);
            }
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 af[MT], bf[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (MIXED && !A_KS) af[i] = TA::frag_as_ks(la, wr * MT + i, s, lane);
                else af[i] = TA::frag(la, wr * MT + i, s, lane);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (MIXED && !B_KS) bf[j] = TB::frag_as_ks(lb, wc * 2 + j, s, lane);
                else bf[j] = TB::frag(lb, wc * 2 + j, s, lane);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = bn + (wc * 2 + j) * 32 + l31;
            if (n >= g.N) continue;
            const float bv = (g.splits <= 1 && g.bias) ? g.bias[n] : 0.f;
            // the tile's 16 residual values are requested TOGETHER (clamped row, no branch per element) and waited for once:
            // `if (g.residual) v += g.residual[..]` inside the loop put every load, its s_waitcnt vmcnt(0) and the add into a
            // block of their own -- 16 x MT x 2 serialised L2 round trips per lane at the end of every workgroup
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            if (g.splits <= 1 && g.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int64_t m = bm + (wr * MT + i) * 32 + mfma32_row(r, hf);
                    m = m < g.M ? m : g.M - 1;
                    rv[r] = g.residual[m * g.ldr + n];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = bm + (wr * MT + i) * 32 + mfma32_row(r, hf);
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (g.splits > 1) {
                    ((float*)g.C)[((int64_t)bz * g.M + m) * g.N + n] = v;
                } else {
                    v += bv;
                    if (g.preact) g.preact[m * g.ldc + n] = v;
                    v = act_apply(v, g.act);
                    v += rv[r];
                    if constexpr (C16) ((bf16_t*)g.C)[m * g.ldc + n] = __builtin_bit_cast(bf16_t, (__bf16)v);
                    else ((float*)g.C)[m * g.ldc + n] = v;
                }
            }
        }
}

template <bool A_KS, bool B_KS, int BM, bool VEC, int OCC, bool A16 = false, bool B16 = false, bool C16 = false>
int launch_bm(const GArgs& g, int a_vec, int b_vec, int splits, hipStream_t st) {
    const size_t lds = Tile<A_KS, BM>::BYTES + Tile<B_KS, BN>::BYTES;
    auto kern = k_gemm_bf16<A_KS, B_KS, BM, VEC, OCC, A16, B16, C16>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            gaot_set_error("gemm_bf16: cannot set dynamic LDS %zu: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(g.M, BM), (unsigned)ceil_div(g.N, BN), (unsigned)splits);
    if (A_KS) grid = dim3((unsigned)(ceil_div(g.M, BM) * ceil_div(g.N, BN) * splits), 1, 1);
    GAOT_KLAUNCH(kern, grid, dim3(256), lds, st, g, a_vec, b_vec);
    return GAOT_OK;
}

static bool dw_bm128() {
    static const bool on = getenv("GAOT_DW_BM128") && atoi(getenv("GAOT_DW_BM128")) != 0;
    return on;
}

template <bool A_KS, bool B_KS>
int launch(const GArgs& g, int a_vec, int b_vec, int splits, int dt, hipStream_t st) {
    // fewer than two 128x128 workgroups per CU: halve the tile height so that twice as many workgroups hide latency
    const int64_t blocks128 = ceil_div(g.N, BN) * ceil_div(g.M, 128) * splits;
    const bool small = blocks128 < 2100 && g.M > 64;   // measured: 64-row tiles win up to ~8 workgroups per CU
    // buffer-addressed tiles need 16-byte aligned rows, whole 16-byte chunks along k and 31-bit byte offsets inside a tile
    const int64_t span_a = (A_KS ? g.K : 128) * g.lda * 4, span_b = (B_KS ? g.K : 128) * g.ldb * 4;
    const bool k4 = (!A_KS && !(dt & 1)) || (!B_KS && !(dt & 2));   // an fp32 operand read in float4s along k
    const bool vec = a_vec && b_vec && (!k4 || g.K % 4 == 0) && span_a < 0x7fffffff && span_b < 0x7fffffff;
    if (dt) {   // some operand / the result is bf16 in memory (dt bits: 1 = A, 2 = B, 4 = C): 64-row tiles only
        const bool k8 = ((dt & 1) && !A_KS) || ((dt & 2) && !B_KS);   // a bf16 operand whose 16-byte chunks run along k
        if (!vec || (k8 && g.K % 8 != 0) || ((dt & 4) && splits > 1)) {
            gaot_set_error("gemm: bf16 operands need 16-byte aligned rows, K %% 8 == 0 along contiguous k, and no split-K result");
            return GAOT_ERR_UNSUPPORTED;
        }
        switch (dt) {
            case 1: return launch_bm<A_KS, B_KS, 64, true, 2, true, false, false>(g, a_vec, b_vec, splits, st);
            case 2: return launch_bm<A_KS, B_KS, 64, true, 2, false, true, false>(g, a_vec, b_vec, splits, st);
            case 3:
                // lab switch (GAOT_DW_BM128=1): 128-row tiles for the split-K weight gradients with both operands bf16 -- the B panel
                // (x: [K][N]) is re-read once per row block of the output, so twice the tile height halves that share of the L2 traffic
                if (A_KS && B_KS && g.M >= 256 && dw_bm128()) return launch_bm<A_KS, B_KS, 128, true, 2, true, true, false>(g, a_vec, b_vec, splits, st);
                return launch_bm<A_KS, B_KS, 64, true, 2, true, true, false>(g, a_vec, b_vec, splits, st);
            case 4: return launch_bm<A_KS, B_KS, 64, true, 2, false, false, true>(g, a_vec, b_vec, splits, st);
            case 5: return launch_bm<A_KS, B_KS, 64, true, 2, true, false, true>(g, a_vec, b_vec, splits, st);
            case 6: return launch_bm<A_KS, B_KS, 64, true, 2, false, true, true>(g, a_vec, b_vec, splits, st);
            case 7: return launch_bm<A_KS, B_KS, 64, true, 2, true, true, true>(g, a_vec, b_vec, splits, st);
            default:
                gaot_set_error("gemm: unsupported bf16 operand combination %d", dt);
                return GAOT_ERR_UNSUPPORTED;
        }
    }
    if (vec) {
        if (small) return launch_bm<A_KS, B_KS, 64, true, 2>(g, a_vec, b_vec, splits, st);
        return launch_bm<A_KS, B_KS, 128, true, 2>(g, a_vec, b_vec, splits, st);
    }
    if (small) return launch_bm<A_KS, B_KS, 64, false, 2>(g, a_vec, b_vec, splits, st);
    return launch_bm<A_KS, B_KS, 128, false, 2>(g, a_vec, b_vec, splits, st);
}

}  // namespace

bool gaot_gemm_k256_applicable(const void* A, const void* W, const void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                               int64_t ldc, int c16);
int gaot_gemm_k256_launch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t lda, int64_t ldw, int64_t ldc,
                          int c16, hipStream_t st);
bool gaot_gemm_tn_n256_applicable(const void* A, const void* W, const void* C, const void* R, int64_t M, int64_t N, int64_t K, int64_t lda,
                                  int64_t ldw, int64_t ldc, int64_t ldr);
int gaot_gemm_tn_n256_launch(const void* A, const void* W, float* C, const float* R, int64_t M, int64_t K, int64_t lda, int64_t ldw,
                             int64_t ldc, int64_t ldr, hipStream_t st);

// called by gaot_gemm / gaot_gemm_ex (gemm.hip) for precision == 1 when the output is wide enough for the 128-wide tile.
// dt: bit 0 = A is bf16 in memory, bit 1 = B, bit 2 = C (leading dimensions always in elements).
int gaot_gemm_bf16_dispatch(const void* A, const void* B, void* C, float* preact, const float* bias,
                            const float* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                            int64_t ldc, int64_t ldr, int a_trans, int b_trans, int act, int splits, int64_t kps,
                            int a_vec, int b_vec, int dt, hipStream_t st) {
    GArgs g{A, B, C, preact, bias, residual, M, N, K, lda, ldb, ldc, ldr, act, splits, kps};
    const bool a_ks = a_trans != 0;   // A(m,k) = A[k*lda + m]
    const bool b_ks = b_trans == 0;   // B(k,n) = B[k*ldb + n]
    // x W^T with K = 256, both operands bf16 in memory and a bare epilogue (q|k|v and w1|w3 forward): weights-in-registers kernel
    if (!a_ks && !b_ks && (dt & 3) == 3 && splits <= 1 && !bias && !residual && !preact && act == 0 &&
        gaot_gemm_k256_applicable(A, B, C, M, N, K, lda, ldb, ldc, (dt & 4) != 0))
        return gaot_gemm_k256_launch(A, B, C, M, N, lda, ldb, ldc, (dt & 4) != 0, st);
    // x W^T with N = 256 and a long K, both operands bf16 in memory, fp32 result (+ residual): streamed-weight kernel
    if (!a_ks && !b_ks && dt == 3 && splits <= 1 && !bias && !preact && act == 0 &&
        gaot_gemm_tn_n256_applicable(A, B, C, residual, M, N, K, lda, ldb, ldc, ldr))
        return gaot_gemm_tn_n256_launch(A, B, (float*)C, residual, M, K, lda, ldb, ldc, ldr, st);
    // (measured and removed, round 4, profiles/archive/r4_p_wgrad_streamed_operand_lab.txt: a weight-gradient kernel with 128 x 128 tiles whose
    // token-major bf16 operands stream by LDS-DMA through a ring of 3-4 stage buffers (source-side XOR swizzle for the transposed
    // reads) -- 60 / 56 us against 52 us for the register-staged kernel below on w1|w3 with operands from HBM; its DMA alone
    // takes 38 us (2.3 TB/s: 64-96 KB in flight per CU is all an LDS ring can hold) and one wave per SIMD does not overlap
    // its fragment reads with its MFMAs)
    if (!a_ks && !b_ks) return launch<false, false>(g, a_vec, b_vec, splits, dt, st);
    if (!a_ks && b_ks) return launch<false, true>(g, a_vec, b_vec, splits, dt, st);
    if (a_ks && !b_ks) return launch<true, false>(g, a_vec, b_vec, splits, dt, st);
    return launch<true, true>(g, a_vec, b_vec, splits, dt, st);
}
