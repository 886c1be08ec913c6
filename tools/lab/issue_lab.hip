// Issue-rate lab: how do MFMA and VALU streams share one SIMD?  (standalone; hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int LB>
__global__ __launch_bounds__(256, LB) void k_lab(float* out, int iters, float seed) {
    // MODE 0: 8 MFMA/iter (2 chains)   1: VALU only (16 exp, 16 mul, 8 cvt-like)   2: both, independent   3: both, dependent
    f32x16 a0, a1, v;
    bf16x8 x, y;
    for (int r = 0; r < 16; ++r) { a0[r] = seed * r; a1[r] = seed + r; v[r] = seed * 0.001f * (r + threadIdx.x); }
    for (int j = 0; j < 8; ++j) { x[j] = (short)(threadIdx.x + j); y[j] = (short)(threadIdx.x * 3 + j); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(v[r]);
                v[r] = p * 0.999f - 0.5f;
            }
        }
        if (MODE == 3) {
            // dependent: MFMA -> exp/mul on its result -> pack -> MFMA
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            bf16x8 p0, p1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(a0[r] * 1e-30f);
                a0[r] = p;
                a1[r] = p * a1[r] * 1e-30f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0[j] = (short)__builtin_bit_cast(unsigned short, (__bf16)a0[j]);
                p1[j] = (short)__builtin_bit_cast(unsigned short, (__bf16)a1[j + 8]);
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, x, a1, 0, 0, 0);
        }
    }
    if (MODE == 7) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        f32x16 dk = a0, dv = a1, c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = -100.f - r; c1[r] = 0.001f * r; }
        for (int it = 0; it < iters; ++it) {
            f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c1, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, dp, 0, 0, 0);
            u4 pw0, pw1, dw0, dw1;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float pa = __builtin_amdgcn_exp2f(sc[2 * g]), pb = __builtin_amdgcn_exp2f(sc[2 * g + 1]);
                const f2 pv = {pa, pb};
                const f2 dvv = {pa * dp[2 * g], pb * dp[2 * g + 1]};
                const unsigned pw = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, b2));
                const unsigned dw = __builtin_bit_cast(unsigned, __builtin_convertvector(dvv, b2));
                if (g < 4) { pw0[g & 3] = pw; dw0[g & 3] = dw; } else { pw1[g & 3] = pw; dw1[g & 3] = dw; }
            }
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, __builtin_bit_cast(bf16x8, pw0), dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, __builtin_bit_cast(bf16x8, dw0), dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, __builtin_bit_cast(bf16x8, pw1), dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, __builtin_bit_cast(bf16x8, dw1), dk, 0, 0, 0);
        }
        a0 = dk; a1 = dv;
    }
    if (MODE == 8) {
        // forward-attention unit: 2 MFMA (S) -> max over 16 + cross-half + branch -> 16 exp, 16 add, 8 cvt_pk -> 2 MFMA (PV)
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        f32x16 acc = a0, negm;
        for (int r = 0; r < 16; ++r) negm[r] = -50.f;
        float l = 0.f, m = 50.f;
        for (int it = 0; it < iters; ++it) {
            f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, negm, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, sc, 0, 0, 0);
            float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
            mx = fmaxf(mx, sc[15]);
            const unsigned u = __builtin_bit_cast(unsigned, mx);
            const auto rr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)rr[0]), __builtin_bit_cast(float, (unsigned)rr[1]));
            if (__any(mx > 1e30f)) {
                m += mx;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] -= mx; acc[r] *= 0.5f; negm[r] = -m; }
            }
            float ps0 = 0.f, ps1 = 0.f;
            u4 pw0, pw1;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float pa = __builtin_amdgcn_exp2f(sc[2 * g] * 1e-30f), pb = __builtin_amdgcn_exp2f(sc[2 * g + 1] * 1e-30f);
                ps0 += pa; ps1 += pb;
                const f2 pv = {pa, pb};
                const unsigned pw = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, b2));
                if (g < 4) pw0[g & 3] = pw; else pw1[g & 3] = pw;
            }
            l += ps0 + ps1;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, __builtin_bit_cast(bf16x8, pw0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, __builtin_bit_cast(bf16x8, pw1), acc, 0, 0, 0);
        }
        a0 = acc; a1[0] = l + m;
    }
    if (MODE == 4) {
        // software-pipelined + hand-interleaved: per unit 8 slots of [1 MFMA | 2 exp, 2 mul, 2 cvt_pk]; the MFMAs of a
        // slot never depend on the VALU work beside them (S/dP of the NEXT unit, dV/dK of the previous half)
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        f32x16 sc = a0, dp = a1, scn = a0, dpn = a1, dk = a0, dv = a1;
        u4 pw0 = {1, 2, 3, 4}, dw0 = {1, 2, 3, 4}, pw1 = {1, 2, 3, 4}, dw1 = {1, 2, 3, 4}, pw1p = pw1, dw1p = dw1;
        for (int it = 0; it < iters; ++it) {
#define VAL(g)                                                                                         \
    {                                                                                                  \
        const float pa = __builtin_amdgcn_exp2f(sc[2 * g] * 1e-30f), pb = __builtin_amdgcn_exp2f(sc[2 * g + 1] * 1e-30f);    \
        const f2 pv = {pa, pb};                                                                        \
        const f2 dvv = {pa * dp[2 * g], pb * dp[2 * g + 1]};                                           \
        const unsigned pw = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, b2));             \
        const unsigned dw = __builtin_bit_cast(unsigned, __builtin_convertvector(dvv, b2));            \
        if (g < 4) { pw0[g & 3] = pw; dw0[g & 3] = dw; } else { pw1[g & 3] = pw; dw1[g & 3] = dw; }    \
    }
#define F() __builtin_amdgcn_sched_barrier(0)
            F(); dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, __builtin_bit_cast(bf16x8, pw1p), dv, 0, 0, 0); F(); VAL(0) F();
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, __builtin_bit_cast(bf16x8, dw1p), dk, 0, 0, 0); F(); VAL(1) F();
            scn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, scn, 0, 0, 0); F(); VAL(2) F();
            dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, dpn, 0, 0, 0); F(); VAL(3) F();
            scn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, scn, 0, 0, 0); F(); VAL(4) F();
            dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, dpn, 0, 0, 0); F(); VAL(5) F();
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, __builtin_bit_cast(bf16x8, pw0), dv, 0, 0, 0); F(); VAL(6) F();
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, __builtin_bit_cast(bf16x8, dw0), dk, 0, 0, 0); F(); VAL(7) F();
            pw1p = pw1; dw1p = dw1;
            sc = scn; dp = dpn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { scn[r] = seed; dpn[r] = seed; }
        }
        a0 = dk; a1 = dv;
        for (int r = 0; r < 16; ++r) v[r] = sc[r] + dp[r];
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ping-pong: 512-thread workgroup, waves 0-3 and 4-7 share the SIMDs pairwise; one half runs the MFMA phase while the
// other runs the VALU phase of a dependent chain, s_barrier between phases
__global__ __launch_bounds__(512, 1) void k_pingpong(float* out, int iters, float seed) {
    f32x16 a0, a1;
    bf16x8 x, y;
    for (int r = 0; r < 16; ++r) { a0[r] = seed * r; a1[r] = seed + r; }
    for (int j = 0; j < 8; ++j) { x[j] = (short)(threadIdx.x + j); y[j] = (short)(threadIdx.x * 3 + j); }
    const bool second = threadIdx.x >= 256;
    bf16x8 p0 = x, p1 = y;
    if (second) __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // MFMA phase: dV/dK of the previous unit (needs p0,p1) + S/dP of this unit
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        __syncthreads();
        // VALU phase on the results
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(a0[r] * 1e-30f);
            a0[r] = p;
            a1[r] = p * a1[r] * 1e-30f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p0[j] = (short)__builtin_bit_cast(unsigned short, (__bf16)a0[j]);
            p1[j] = (short)__builtin_bit_cast(unsigned short, (__bf16)a1[j + 8]);
        }
        __syncthreads();
    }
    if (!second) __syncthreads();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s + p0[0];
}

template <int MODE, int LB = 1>
void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = 20000;
    size_t pad = wgs_per_cu == 1 ? 140000 : (wgs_per_cu == 2 ? 70000 : (wgs_per_cu == 4 ? 36000 : 0));
    hipFuncSetAttribute((const void*)k_lab<MODE, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k_lab<MODE, LB>), dim3(grid), dim3(256), pad, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_lab<MODE, LB>), dim3(grid), dim3(256), pad, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD runs wgs_per_cu waves; ns per iteration per SIMD-wave-slot
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.1f ns per iteration per wave, %.1f ns per iteration of SIMD throughput\n", name,
           wgs_per_cu, ms, ms * 1e6 / iters, ms * 1e6 / iters / wgs_per_cu);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("8 MFMA", w, out);
        run<1>("16 exp+16 fma", w, out);
        run<2>("8 MFMA + VALU independent", w, out);
        run<3>("8 MFMA + VALU dependent", w, out);
        run<4>("8 MFMA + VALU pipelined+interleaved", w, out);
        run<2, 2>("independent, VGPR accumulators", w, out);
        run<7, 4>("real dkv mix, dependent, no memory", w, out);
        run<8, 4>("fwd-attention mix, dependent, no memory", w, out);
        run<4, 2>("pipelined+interleaved, VGPR acc", w, out);
    }
    {
        const int iters = 20000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(512), 0, 0, out, 100, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ping-pong 2 waves/SIMD (dependent chain, anti-phase): %.3f ms -> %.1f ns per iteration per wave, %.1f ns per unit of SIMD throughput\n",
               ms, ms * 1e6 / iters, ms * 1e6 / iters / 2);
    }
    return 0;
}
