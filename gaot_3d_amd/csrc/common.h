// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the GAOT-3D hot path.
// Wave = 64 lanes everywhere in this library; nothing here is portable to 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/gaot3d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

void gaot_set_error(const char* fmt, ...);

// every kernel launch of the library goes through this macro: a process-wide launch counter (gaot_launch_count) lets the
// host report launches per step without a profiler
extern long long g_gaot_launches;
#define GAOT_KLAUNCH(...)                \
    do {                                 \
        ++g_gaot_launches;               \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define GAOT_CHECK_ARG(cond, msg)                         \
    do {                                                  \
        if (!(cond)) {                                    \
            gaot_set_error("%s: %s", __func__, msg);      \
            return GAOT_ERR_ARG;                          \
        }                                                 \
    } while (0)

// clear any sticky error left by an earlier failed call of this thread
#define GAOT_ENTER() (void)hipGetLastError()

#define GAOT_LAUNCH_CHECK()                                                        \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            gaot_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return GAOT_ERR_LAUNCH;                                                \
        }                                                                          \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row of a 32x32 MFMA C/D tile held in register r (0..15) of a lane in half h (lane>>5):
// row = (r&3) + 8*(r>>2) + 4*h ; column = lane&31.   (dtype independent on gfx950)
__device__ __forceinline__ constexpr int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// exact-erf GELU (torch F.gelu default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Fast erf-GELU for the per-edge MLP (192 activations per edge make erff the top VALU cost of the GNO kernels):
// Abramowitz-Stegun 7.1.26 on u = |x|/sqrt(2) -- 1 v_rcp + 1 v_exp + 7 FMA, the exponential is shared with
// gelu'(x).  Measured max |error| over [-8, 8] in fp32: 4.2e-7 (gelu), 3.2e-7 (gelu') -- the level of fp32 erff.
__device__ __forceinline__ void gelu_fast_pair(float x, float& g, float& dg) {
    const float u = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2)
    float p = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);              // the 1/2 of Phi folded into the coefficients
    p = fmaf(t, p, 0.5f * 1.421413741f);
    p = fmaf(t, p, 0.5f * -0.284496736f);
    p = fmaf(t, p, 0.5f * 0.254829592f);
    const float h = t * p * e;                    // Phi(-|x|)
    const float cdf = x >= 0.f ? 1.0f - h : h;
    g = x * cdf;
    dg = fmaf(x * e, 0.39894228040143267794f, cdf);
}
__device__ __forceinline__ float gelu_fast(float x) {   // same function, forward only: no Phi(x) select
    const float u = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    float p = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    p = fmaf(t, p, 0.5f * 1.421413741f);
    p = fmaf(t, p, 0.5f * -0.284496736f);
    p = fmaf(t, p, 0.5f * 0.254829592f);
    return fmaf(-fabsf(x), t * p * e, fmaxf(x, 0.f));   // x Phi(x) = max(x, 0) - |x| Phi(-|x|)
}

// Two activations at a time on the packed-fp32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 run two lanes' worth of fp32 per
// issue slot): the GNO kernels are bound by this arithmetic (192 activations per edge), and left to itself the compiler
// packs only part of the scalar form.  Same approximation as gelu_fast / gelu_fast_pair, regrouped so that no select is
// needed:  x Phi(x) = x/2 + |x| (1/2 - Phi(-|x|)),   Phi(x) = 1/2 + sign(x) (1/2 - Phi(-|x|)).
typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32v2 gelu_half_minus_tail2(f32v2 x, f32v2 ax, f32v2& e) {   // 1/2 - Phi(-|x|), e = exp(-x^2/2)
    const f32v2 d = __builtin_elementwise_fma(ax, (f32v2)(0.3275911f * 0.70710678118654752440f), (f32v2)(1.0f));
    const f32v2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32v2 s = x * (f32v2)(0.84932180028801904272f);       // sqrt(log2(e) / 2): exp2(-(s*s)) = exp(-x^2/2)
    const f32v2 q = s * s;
    e = f32v2{__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    f32v2 p = __builtin_elementwise_fma(t, (f32v2)(0.5f * 1.061405429f), (f32v2)(0.5f * -1.453152027f));
    p = __builtin_elementwise_fma(t, p, (f32v2)(0.5f * 1.421413741f));
    p = __builtin_elementwise_fma(t, p, (f32v2)(0.5f * -0.284496736f));
    p = __builtin_elementwise_fma(t, p, (f32v2)(0.5f * 0.254829592f));
    const f32v2 tp = t * p;
    return __builtin_elementwise_fma(-tp, e, (f32v2)(0.5f));
}
__device__ __forceinline__ f32v2 gelu_fast2(f32v2 x) {
    const f32v2 ax = {fabsf(x[0]), fabsf(x[1])};
    f32v2 e;
    const f32v2 w = gelu_half_minus_tail2(x, ax, e);
    return __builtin_elementwise_fma(ax, w, x * (f32v2)(0.5f));
}
__device__ __forceinline__ void gelu_fast_pair2(f32v2 x, f32v2& g, f32v2& dg) {
    const f32v2 ax = {fabsf(x[0]), fabsf(x[1])};
    f32v2 e;
    const f32v2 w = gelu_half_minus_tail2(x, ax, e);
    g = __builtin_elementwise_fma(ax, w, x * (f32v2)(0.5f));
    const f32v2 sw = {copysignf(w[0], x[0]), copysignf(w[1], x[1])};
    const f32v2 cdf = sw + (f32v2)(0.5f);
    dg = __builtin_elementwise_fma(x * e, (f32v2)(0.39894228040143267794f), cdf);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
