// Dense GEMM on the gfx950 matrix cores for the per-node linears and the Transformer projections:
//     C[m][n] = act( sum_k A(m,k) * B(k,n) + bias[n] ) + residual[m][n]
// Stands in for ATen addmm/mm behind nn.Linear (reference src/model/layers/mlp.py:327-335,
// attn.py:104-106,129,156, gaot_3d.py:205, magno.py:543,573,794) and for their autograd:
//   forward        y  = x W^T + b      : a_trans=0, b_trans=1 (W is [N][K], nn.Linear layout)
//   grad input     dx = dy W           : a_trans=0, b_trans=0
//   grad weight    dW = dy^T x         : a_trans=1, b_trans=0, split over the (long) row dimension
//
// precision 0: v_mfma_f32_32x32x2_f32  -- exact fp32 products, fp32 accumulate (parity mode)
// precision 1: v_mfma_f32_32x32x16_bf16 -- operands rounded to bf16 when staged, fp32 accumulate
//
// Tiling: 256 threads = 4 waves; each wave owns MT x NT tiles of 32x32; operands are staged in LDS
// in whichever orientation keeps the GLOBAL read coalesced (k-contiguous rows or m-contiguous
// columns); both operands walk k in the same permuted order so no transpose is ever needed.
#include "common.h"

namespace {

constexpr int BK = 16;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    float* preact;        // optional: value before the activation
    const float* bias;    // [N] or null
    const float* residual;
    int64_t M, N, K;      // K = full reduction length
    int64_t lda, ldb, ldc, ldr;
    int act;              // GAOT_ACT_* (common.h): 0 none, 1 gelu(erf), 2 relu, 3 silu, 4.. the wider F.<name> surface
    int splits;           // >1: write raw partials [split][M][N] to C (ldc = N), epilogue done by k_splitk_reduce
    int64_t k_per_split;
};

// the epilogue knows the shipped activations only (a switch over the whole GAOT_ACT_* set would drag libm's tanhf / expm1f
// into every GEMM instantiation); the others run as gaot_act_fwd on the pre-activation (functional.LinearFn)
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return gelu_f(v);
        case 2: return v > 0.f ? v : 0.f;
        case 3: return v / (1.f + __expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ short f2bf(float f) {  // round-to-nearest-even, NaN preserving via cast
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(short, b);
}

// One operand tile [ROWS x BK] staged in LDS.
//   MC == false: source is k-contiguous (X[row*ld + k]);  LDS [ROWS][BK+4], fragments by ds_read_b128
//   MC == true : source is row-contiguous (X[k*ld + row]); LDS [BK][ROWS+4], fragments by ds_read_b32
template <int ROWS, bool MC>
struct OperandTile {
    static constexpr int LDK = BK + 4;
    static constexpr int LDR = ROWS + 4;
    static constexpr int FLOATS = MC ? BK * LDR : ROWS * LDK;
    static constexpr int NVEC = ROWS * BK / 4;                            // float4 per tile
    static constexpr int VEC_PER_THREAD = (NVEC + 255) / 256;            // float4 per thread

    float4 regs[VEC_PER_THREAD];

    __device__ __forceinline__ void load(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                         int64_t kend, bool vec_ok) {
#pragma unroll
        for (int v = 0; v < VEC_PER_THREAD; ++v) {
            const int idx = threadIdx.x + v * 256;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx >= NVEC) {
            } else if (!MC) {
                const int r = idx / (BK / 4), kq = idx % (BK / 4);
                const int64_t row = row0 + r, k = k0 + 4 * kq;
                if (row < nrows) {
                    const float* p = X + row * ld + k;
                    if (vec_ok && k + 3 < kend) {
                        t = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (k + 0 < kend) t.x = p[0];
                        if (k + 1 < kend) t.y = p[1];
                        if (k + 2 < kend) t.z = p[2];
                        if (k + 3 < kend) t.w = p[3];
                    }
                }
            } else {
                const int kk = idx / (ROWS / 4), rq = idx % (ROWS / 4);
                const int64_t k = k0 + kk, row = row0 + 4 * rq;
                if (k < kend) {
                    const float* p = X + k * ld + row;
                    if (vec_ok && row + 3 < nrows) {
                        t = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (row + 0 < nrows) t.x = p[0];
                        if (row + 1 < nrows) t.y = p[1];
                        if (row + 2 < nrows) t.z = p[2];
                        if (row + 3 < nrows) t.w = p[3];
                    }
                }
            }
            regs[v] = t;
        }
    }
    __device__ __forceinline__ void store(float* lds) const {
#pragma unroll
        for (int v = 0; v < VEC_PER_THREAD; ++v) {
            const int idx = threadIdx.x + v * 256;
            if (idx >= NVEC) {
            } else if (!MC) {
                const int r = idx / (BK / 4), kq = idx % (BK / 4);
                *reinterpret_cast<float4*>(lds + r * LDK + 4 * kq) = regs[v];
            } else {
                const int kk = idx / (ROWS / 4), rq = idx % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + kk * LDR + 4 * rq) = regs[v];
            }
        }
    }
    // fragment for tile-row `r` (0..ROWS-1 = 32*tile + lane&31), k-group kg (8 k's): 4 values, j-th
    // value is k = kg*8 + 4*hf + j
    static __device__ __forceinline__ float4 frag(const float* lds, int r, int kg, int hf) {
        if (!MC) {
            return *reinterpret_cast<const float4*>(lds + r * LDK + kg * 8 + 4 * hf);
        } else {
            float4 t;
            const float* p = lds + (kg * 8 + 4 * hf) * LDR + r;
            t.x = p[0];
            t.y = p[LDR];
            t.z = p[2 * LDR];
            t.w = p[3 * LDR];
            return t;
        }
    }
};

template <int WR, int WC, int MT, int NT, bool A_MC, bool B_NC, bool BF16>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g, int a_vec, int b_vec) {
    constexpr int BM = WR * MT * 32, BN = WC * NT * 32;
    static_assert(WR * WC == 4, "4 waves");
    using TA = OperandTile<BM, A_MC>;
    using TB = OperandTile<BN, B_NC>;
    __shared__ __attribute__((aligned(16))) float lds[TA::FLOATS + TB::FLOATS];
    float* la = lds;
    float* lb = lds + TA::FLOATS;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, hf = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const int64_t bm = (int64_t)blockIdx.y * BM, bn = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t kend = (g.splits > 1) ? ((kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K) : g.K;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TA ta;
    TB tb;
    if (kbeg < kend) {
        ta.load(g.A, g.lda, bm, g.M, kbeg, kend, a_vec);
        tb.load(g.B, g.ldb, bn, g.N, kbeg, kend, b_vec);
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        ta.store(la);
        tb.store(lb);
        __syncthreads();
        if (k0 + BK < kend) {
            ta.load(g.A, g.lda, bm, g.M, k0 + BK, kend, a_vec);
            tb.load(g.B, g.ldb, bn, g.N, k0 + BK, kend, b_vec);
        }
        if constexpr (!BF16) {
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = TA::frag(la, (wr * MT + i) * 32 + l31, kg, hf);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = TB::frag(lb, (wc * NT + j) * 32 + l31, kg, hf);
            {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const float a = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
                            const float b = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                        }
            }
        }
        }
        if constexpr (BF16) {
            // one 32x32x16 step per BK: lane half hf supplies k = 8*hf + j (j = 0..7); our fragments hold
            // k = kg*8 + 4*hf + s, i.e. for half hf: kg=0 -> 4hf..4hf+3, kg=1 -> 8+4hf..  Any k order is
            // fine as long as A and B agree, so feed [kg=0 frag | kg=1 frag] on both sides.
            bf16x8 a8[MT], b8[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float4 f0 = TA::frag(la, (wr * MT + i) * 32 + l31, 0, hf);
                const float4 f1 = TA::frag(la, (wr * MT + i) * 32 + l31, 1, hf);
                a8[i][0] = f2bf(f0.x); a8[i][1] = f2bf(f0.y); a8[i][2] = f2bf(f0.z); a8[i][3] = f2bf(f0.w);
                a8[i][4] = f2bf(f1.x); a8[i][5] = f2bf(f1.y); a8[i][6] = f2bf(f1.z); a8[i][7] = f2bf(f1.w);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 f0 = TB::frag(lb, (wc * NT + j) * 32 + l31, 0, hf);
                const float4 f1 = TB::frag(lb, (wc * NT + j) * 32 + l31, 1, hf);
                b8[j][0] = f2bf(f0.x); b8[j][1] = f2bf(f0.y); b8[j][2] = f2bf(f0.z); b8[j][3] = f2bf(f0.w);
                b8[j][4] = f2bf(f1.x); b8[j][5] = f2bf(f1.y); b8[j][6] = f2bf(f1.z); b8[j][7] = f2bf(f1.w);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t n = bn + (wc * NT + j) * 32 + l31;
            if (n >= g.N) continue;
            const float bv = (g.splits <= 1 && g.bias) ? g.bias[n] : 0.f;
            // residual values of the tile requested together, waited for once (see gemm_bf16.hip)
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
                    g.C[((int64_t)blockIdx.z * g.M + m) * g.N + n] = v;
                } else {
                    v += bv;
                    if (g.preact) g.preact[m * g.ldc + n] = v;
                    v = apply_act(v, g.act);
                    v += rv[r];
                    g.C[m * g.ldc + n] = v;
                }
            }
        }
}

// Fixed-order reduction of the split-K partials [split][M*N]: a workgroup owns 64 consecutive outputs; its 4 waves
// each sum every 4th split (coalesced 256-byte rows), then the 4 partial sums are added in wave order -- the order
// never depends on timing, so the result is bit-reproducible.
// OUT outputs x (256 / OUT) split lanes per workgroup: 64 x 4 for the Transformer-sized outputs, 16 x 16 for the small
// per-point weight gradients whose 500 K-row reduction is cut into up to 1024 splits (64 x 4 walked 256 splits serially
// per lane there: 62 us for a 32 x 64 output)
template <int OUT>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, int splits, GemmArgs g) {
    constexpr int LANES = 256 / OUT;
    __shared__ float red[LANES][OUT];
    const int o = threadIdx.x % OUT, sl = threadIdx.x / OUT;
    const int64_t mn = g.M * g.N;
    const int64_t i = (int64_t)blockIdx.x * OUT + o;
    float v = 0.f;
    if (i < mn)
        for (int s = sl; s < splits; s += LANES) v += part[(int64_t)s * mn + i];
    red[sl][o] = v;
    __syncthreads();
    if (sl != 0 || i >= mn) return;
    v = red[0][o];
#pragma unroll
    for (int j = 1; j < LANES; ++j) v += red[j][o];
    const int64_t m = i / g.N, n = i % g.N;
    if (g.bias) v += g.bias[n];
    if (g.preact) g.preact[m * g.ldc + n] = v;
    v = apply_act(v, g.act);
    if (g.residual) v += g.residual[m * g.ldr + n];
    g.C[m * g.ldc + n] = v;
}

template <int WR, int WC, int MT, int NT>
void launch_cfg(const GemmArgs& g, int a_trans, int b_trans, int bf16, int a_vec, int b_vec, int splits, hipStream_t st) {
    constexpr int BM = WR * MT * 32, BN = WC * NT * 32;
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits);
    const bool a_mc = a_trans != 0, b_nc = b_trans == 0;
#define GAOT_GEMM_CASE(AM, BNc, BF)                                                                      \
    if (a_mc == AM && b_nc == BNc && (bf16 != 0) == BF) {                                                \
        GAOT_KLAUNCH((k_gemm<WR, WC, MT, NT, AM, BNc, BF>), grid, dim3(256), 0, st, g, a_vec, b_vec); \
        return;                                                                                          \
    }
    GAOT_GEMM_CASE(false, false, false)
    GAOT_GEMM_CASE(false, true, false)
    GAOT_GEMM_CASE(true, false, false)
    GAOT_GEMM_CASE(true, true, false)
    GAOT_GEMM_CASE(false, false, true)
    GAOT_GEMM_CASE(false, true, true)
    GAOT_GEMM_CASE(true, false, true)
    GAOT_GEMM_CASE(true, true, true)
#undef GAOT_GEMM_CASE
}

}  // namespace

struct GemmPlan {
    int cfg;       // 0: 128x32, 1: 128x64, 2: 128x128
    int splits;
    int64_t kps;
};
int gaot_gemm_bf16_dispatch(const void* A, const void* B, void* C, float* preact, const float* bias,
                            const float* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                            int64_t ldc, int64_t ldr, int a_trans, int b_trans, int act, int splits, int64_t kps,
                            int a_vec, int b_vec, int dt, hipStream_t st);

static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int bk = BK) {
    GemmPlan p;
    p.cfg = N <= 32 ? 0 : (N <= 64 ? 1 : 2);
    const int bm = 128, bn = p.cfg == 0 ? 32 : (p.cfg == 1 ? 64 : 128);
    const int64_t tiles = ceil_div(M, bm) * ceil_div(N, bn);
    int64_t s = 1;
    if (tiles < 128 && K >= 4096) {
        // weight gradients of the per-point MLPs reduce over 500 K rows into ONE tile: a split walks its rows in
        // sequential 16-row steps (each a memory round trip), so such shapes get ~4 workgroups per CU
        if (K >= 131072) {
            s = std::min<int64_t>(1024, std::max<int64_t>(1, 2048 / tiles));
            s = std::min<int64_t>(s, ceil_div(K, 256));
        } else {
            // Transformer weight gradients (K = tokens): aim at ~512 workgroups of 64 x 128 -- measured at K = 16384:
            // 2048 x 256 outputs 65 -> 56 us with 8 splits, 256 x 256 outputs 24 -> 21 us with 32, the others best at 16
            const int64_t tiles64 = ceil_div(M, 64) * ceil_div(N, 128);
            int64_t want = std::max<int64_t>(1, 512 / tiles64);
            s = 4;
            while (s * 2 <= want && s < 32) s *= 2;
            s = std::min<int64_t>(s, std::max<int64_t>(1, K / 512));
            if (const char* e = getenv("GAOT_DW_SPLITS")) {      // lab switch: force the split count of the Transformer weight gradients
                const int64_t f = atoll(e);
                if (f > 0) s = std::min<int64_t>(f, std::max<int64_t>(1, K / 512));
            }
        }
    }
    p.kps = ceil_div(ceil_div(K, s), bk) * bk;
    p.splits = (int)std::max<int64_t>(1, ceil_div(K, std::max<int64_t>(p.kps, 1)));
    if (p.splits <= 1) { p.splits = 1; p.kps = K; }
    return p;
}

extern "C" size_t gaot_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const GemmPlan p = plan_gemm(M, N, K);   // the 64-deep bf16 tiling never needs more splits than this
    return p.splits > 1 ? sizeof(float) * (size_t)(p.splits * M * N) + 64 : 0;
}

static int gemm_impl(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                     int64_t ldc, int a_trans, int b_trans, int dt, const float* bias, int act, const float* residual,
                     int64_t ldr, float* preact, int precision, void* workspace, size_t workspace_bytes,
                     gaot_stream_t stream, int* defer_splits = nullptr, int* defer_lanes = nullptr) {
    if (defer_splits) { *defer_splits = 1; *defer_lanes = 4; }
    GAOT_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    GAOT_CHECK_ARG(act >= 0 && act <= 3, "bad activation id (the epilogue has none / gelu / relu / silu; others: gaot_act_fwd)");
    GAOT_CHECK_ARG(precision == 0 || precision == 1, "precision must be 0 (fp32) or 1 (bf16 operands)");
    if (M == 0 || N == 0) return GAOT_OK;
    GAOT_CHECK_ARG(A && B && C, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int ea = (dt & 1) ? 8 : 4, eb = (dt & 2) ? 8 : 4;   // elements per 16-byte chunk
    const int a_vec = (lda % ea == 0) && (((uintptr_t)A & 15) == 0);
    const int b_vec = (ldb % eb == 0) && (((uintptr_t)B & 15) == 0);
    const bool wide_bf16 = precision == 1 && N > 64;
    if (dt && !wide_bf16) {
        gaot_set_error("gemm: bf16 operands in memory are implemented by the bf16 matrix-core kernel only (precision 1, N > 64)");
        return GAOT_ERR_UNSUPPORTED;
    }
    GemmArgs g;
    g.A = (const float*)A; g.B = (const float*)B; g.C = (float*)C; g.preact = preact; g.bias = bias; g.residual = residual;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.act = act; g.splits = 1; g.k_per_split = K;

    const GemmPlan p = plan_gemm(M, N, K, wide_bf16 ? 64 : BK);
    GemmArgs gk = g;
    float* part = (float*)workspace;
    if (p.splits > 1) {
        GAOT_CHECK_ARG(workspace && workspace_bytes >= gaot_gemm_workspace_bytes(M, N, K), "split-K workspace too small");
        gk.splits = p.splits;
        gk.k_per_split = p.kps;
        gk.C = part;
    }
    if (wide_bf16) {
        int rc = gaot_gemm_bf16_dispatch(A, B, gk.splits > 1 ? (void*)part : C, preact, bias, residual, M, N, K, lda, ldb, ldc,
                                         ldr, a_trans, b_trans, act, gk.splits, gk.k_per_split, a_vec, b_vec, dt, st);
        if (rc != GAOT_OK) return rc;
    } else if (p.cfg == 0) launch_cfg<4, 1, 1, 1>(gk, a_trans, b_trans, precision, a_vec, b_vec, gk.splits, st);
    else if (p.cfg == 1) launch_cfg<4, 1, 1, 2>(gk, a_trans, b_trans, precision, a_vec, b_vec, gk.splits, st);
    else launch_cfg<2, 2, 2, 2>(gk, a_trans, b_trans, precision, a_vec, b_vec, gk.splits, st);
    if (gk.splits > 1 && defer_splits) {      // the caller sums the partials later (gaot_reduce_multi), in the order k_splitk_reduce would
        *defer_splits = gk.splits;
        *defer_lanes = (gk.splits >= 64 && M * N <= 32768) ? 16 : 4;
    } else if (gk.splits > 1) {
        if (gk.splits >= 64 && M * N <= 32768)
            GAOT_KLAUNCH(k_splitk_reduce<16>, dim3((unsigned)ceil_div(M * N, 16)), dim3(256), 0, st, part, gk.splits, g);
        else
            GAOT_KLAUNCH(k_splitk_reduce<64>, dim3((unsigned)ceil_div(M * N, 64)), dim3(256), 0, st, part, gk.splits, g);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_gemm(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldb, int64_t ldc, int a_trans, int b_trans, const float* bias, int act,
                         const float* residual, int64_t ldr, float* preact, int precision, void* workspace,
                         size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    return gemm_impl(A, B, C, M, N, K, lda, ldb, ldc, a_trans, b_trans, 0, bias, act, residual, ldr, preact, precision,
                     workspace, workspace_bytes, stream);
}

extern "C" int gaot_gemm_ex(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int a_trans, int b_trans, int a_bf16, int b_bf16, int c_bf16,
                            const float* bias, int act, const float* residual, int64_t ldr, float* preact, int precision,
                            void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    const int dt = (a_bf16 ? 1 : 0) | (b_bf16 ? 2 : 0) | (c_bf16 ? 4 : 0);
    return gemm_impl(A, B, C, M, N, K, lda, ldb, ldc, a_trans, b_trans, dt, bias, act, residual, ldr, preact, precision,
                     workspace, workspace_bytes, stream);
}

// gaot_gemm_ex (no epilogue) whose split-K completion is LEFT TO THE CALLER: when the plan splits the reduction, only the partial
// products are written ([splits][M*N] fp32 at `workspace`, which the caller keeps) and *splits_out > 1; gaot_reduce_multi sums them
// later -- with *lanes_out -- in exactly the order the in-call pass uses (bit-identical results).  *splits_out == 1: C is complete.
// Weight gradients are read by nobody before the optimizer step: a training step hands all of them to ONE reduction launch.
extern "C" int gaot_gemm_ex_partials(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                     int64_t ldb, int64_t ldc, int a_trans, int b_trans, int a_bf16, int b_bf16, int precision,
                                     void* workspace, size_t workspace_bytes, int* splits_out, int* lanes_out,
                                     gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(splits_out && lanes_out, "null pointer");
    GAOT_CHECK_ARG(ldc == N, "the deferred completion writes a dense [M][N] result");
    const int dt = (a_bf16 ? 1 : 0) | (b_bf16 ? 2 : 0);
    return gemm_impl(A, B, C, M, N, K, lda, ldb, ldc, a_trans, b_trans, dt, nullptr, 0, nullptr, 0, nullptr, precision, workspace,
                     workspace_bytes, stream, splits_out, lanes_out);
}
