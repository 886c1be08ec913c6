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

// Activation ids of the GEMM epilogues and gaot_act_bwd.  0-3 are the shipped configurations' (mlp.py:330-331 F.gelu,
// geoembed.py:37 ReLU, attn.py:156 SiLU); 4.. cover the rest of the reference's `activation_fn(name)` surface
// (src/model/layers/mlp.py:27-35: any F.<name>) with torch's default parameters.
enum { GAOT_ACT_NONE = 0, GAOT_ACT_GELU = 1, GAOT_ACT_RELU = 2, GAOT_ACT_SILU = 3, GAOT_ACT_TANH = 4, GAOT_ACT_LEAKY_RELU = 5,
       GAOT_ACT_ELU = 6, GAOT_ACT_SIGMOID = 7, GAOT_ACT_SOFTPLUS = 8, GAOT_ACT_SELU = 9, GAOT_ACT_RELU6 = 10,
       GAOT_ACT_HARDSWISH = 11, GAOT_ACT_MISH = 12, GAOT_ACT_GELU_TANH = 13, GAOT_ACT_COUNT = 14 };
__device__ __forceinline__ float gaot_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }   // F.softplus threshold 20
__device__ __forceinline__ float gaot_act_fwd(float v, int act) {
    switch (act) {
        case GAOT_ACT_GELU: return gelu_f(v);
        case GAOT_ACT_RELU: return v > 0.f ? v : 0.f;
        case GAOT_ACT_SILU: return v / (1.f + __expf(-v));
        case GAOT_ACT_TANH: return tanhf(v);
        case GAOT_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
        case GAOT_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case GAOT_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case GAOT_ACT_SOFTPLUS: return gaot_softplus(v);
        case GAOT_ACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * expm1f(v));
        case GAOT_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case GAOT_ACT_HARDSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case GAOT_ACT_MISH: return v * tanhf(gaot_softplus(v));
        case GAOT_ACT_GELU_TANH: { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); return 0.5f * v * (1.f + tanhf(u)); }
        default: return v;
    }
}
// d act(v) / dv  (torch autograd's formulas)
__device__ __forceinline__ float gaot_act_grad(float v, int act) {
    switch (act) {
        case GAOT_ACT_GELU: return gelu_grad_f(v);
        case GAOT_ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case GAOT_ACT_SILU: { const float s = 1.f / (1.f + __expf(-v)); return s * (1.f + v * (1.f - s)); }
        case GAOT_ACT_TANH: { const float t = tanhf(v); return 1.f - t * t; }
        case GAOT_ACT_LEAKY_RELU: return v > 0.f ? 1.f : 0.01f;
        case GAOT_ACT_ELU: return v > 0.f ? 1.f : expf(v);
        case GAOT_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-v)); return s * (1.f - s); }
        case GAOT_ACT_SOFTPLUS: return v > 20.f ? 1.f : 1.f / (1.f + expf(-v));
        case GAOT_ACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? 1.f : 1.6732632423543772848170429916717f * expf(v));
        case GAOT_ACT_RELU6: return (v > 0.f && v < 6.f) ? 1.f : 0.f;
        case GAOT_ACT_HARDSWISH: return v < -3.f ? 0.f : (v > 3.f ? 1.f : (2.f * v + 3.f) * (1.f / 6.f));
        case GAOT_ACT_MISH: {
            const float sp = gaot_softplus(v), t = tanhf(sp), sg = 1.f / (1.f + expf(-v));
            return t + v * (1.f - t * t) * sg;
        }
        case GAOT_ACT_GELU_TANH: {
            const float k = 0.7978845608028654f, u = k * (v + 0.044715f * v * v * v), t = tanhf(u);
            return 0.5f * (1.f + t) + 0.5f * v * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * v * v);
        }
        default: return 1.f;
    }
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

// bf16-mode GELU / GELU' (GNO edge MLPs, projection MLP): ONE exponential and no reciprocal per activation.
// log2 Phi(-a) is nearly a polynomial in a (it behaves like -a^2 log2(e)/2 minus a slowly varying term), so
//     Phi(-a) = exp2(P(a)),  P of degree 4 fitted on a = min(|x|, 5.5)  (beyond 5.5: a Phi(-a) < 6e-7),
//     gelu(x) = x Phi(x) = max(x, 0) - a Phi(-a),
//     gelu'(x) = 1/2 + sign(x) (1/2 - Phi(-a) Q(a)),  Q(a) = 1 + ln2 a P'(a)  -- the EXACT derivative of the function above,
// so the backward differentiates what the forward computes.  Max |error| against erf-GELU in fp32 arithmetic over [-12, 12]:
// 9.3e-6 (gelu), 3.7e-5 (gelu'), both far below the bf16 / f16 rounding (2e-3 relative) the results get right after.
// Cost per PAIR of activations on the packed-fp32 VALU forms: 52 issue cycles forward, 84 with the derivative (the
// Abramowitz-Stegun form above: 84 / 104 -- a v_rcp_f32 and four more Horner steps).  Coefficients: tools/fit_gelu.py.
typedef float f32v2 __attribute__((ext_vector_type(2)));
constexpr float GELU_A_MAX = 5.5f;
constexpr float GELU_P0 = -1.000106314f, GELU_P1 = -1.149296311f, GELU_P2 = -0.465028396f, GELU_P3 = -0.04579698f,
                GELU_P4 = 0.004187508f;
constexpr float GELU_Q1 = (float)(0.6931471805599453 * 1.0 * -1.149296311), GELU_Q2 = (float)(0.6931471805599453 * 2.0 * -0.465028396),
                GELU_Q3 = (float)(0.6931471805599453 * 3.0 * -0.04579698), GELU_Q4 = (float)(0.6931471805599453 * 4.0 * 0.004187508);
// min(|x|, GELU_A_MAX) and max(x, 0) in ONE instruction each: v_med3_f32 with the |x| source modifier and an integer max
// on the bit pattern -- fminf / fmaxf cost a second instruction here (the quieting v_max_f32 x, x that IEEE min / max need)
__device__ __forceinline__ float gelu_arg(float x) { return __builtin_amdgcn_fmed3f(fabsf(x), GELU_A_MAX, 0.f); }
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// two activations at a time (v_pk_fma_f32 runs two lanes' worth of fp32 per issue slot; left to itself the compiler packs
// only part of the scalar form)
__device__ __forceinline__ f32v2 gelu_e2_tail2(f32v2 a) {
    f32v2 p = __builtin_elementwise_fma(a, (f32v2)(GELU_P4), (f32v2)(GELU_P3));
    p = __builtin_elementwise_fma(a, p, (f32v2)(GELU_P2));
    p = __builtin_elementwise_fma(a, p, (f32v2)(GELU_P1));
    p = __builtin_elementwise_fma(a, p, (f32v2)(GELU_P0));
    return f32v2{__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
}
__device__ __forceinline__ f32v2 gelu_e2_2(f32v2 x) {
    const f32v2 a = {gelu_arg(x[0]), gelu_arg(x[1])};
    const f32v2 r = {relu_bits(x[0]), relu_bits(x[1])};
    return __builtin_elementwise_fma(-a, gelu_e2_tail2(a), r);
}
__device__ __forceinline__ void gelu_e2_pair2(f32v2 x, f32v2& g, f32v2& dg) {
    const f32v2 a = {gelu_arg(x[0]), gelu_arg(x[1])};
    const f32v2 r = {relu_bits(x[0]), relu_bits(x[1])};
    const f32v2 t = gelu_e2_tail2(a);
    g = __builtin_elementwise_fma(-a, t, r);
    f32v2 q = __builtin_elementwise_fma(a, (f32v2)(GELU_Q4), (f32v2)(GELU_Q3));
    q = __builtin_elementwise_fma(a, q, (f32v2)(GELU_Q2));
    q = __builtin_elementwise_fma(a, q, (f32v2)(GELU_Q1));
    q = __builtin_elementwise_fma(a, q, (f32v2)(1.0f));
    const f32v2 w = __builtin_elementwise_fma(-t, q, (f32v2)(0.5f));
    const f32v2 sw = {copysignf(w[0], x[0]), copysignf(w[1], x[1])};
    dg = sw + (f32v2)(0.5f);
}

__device__ __forceinline__ f32v2 gelu_e2_grad2(f32v2 x) {   // the derivative alone
    const f32v2 a = {gelu_arg(x[0]), gelu_arg(x[1])};
    const f32v2 t = gelu_e2_tail2(a);
    f32v2 q = __builtin_elementwise_fma(a, (f32v2)(GELU_Q4), (f32v2)(GELU_Q3));
    q = __builtin_elementwise_fma(a, q, (f32v2)(GELU_Q2));
    q = __builtin_elementwise_fma(a, q, (f32v2)(GELU_Q1));
    q = __builtin_elementwise_fma(a, q, (f32v2)(1.0f));
    const f32v2 w = __builtin_elementwise_fma(-t, q, (f32v2)(0.5f));
    const f32v2 sw = {copysignf(w[0], x[0]), copysignf(w[1], x[1])};
    return sw + (f32v2)(0.5f);
}

// sigmoid for the bf16-mode SwiGLU (forward and backward, stand-alone passes and GEMM epilogues alike -- ONE form, so the fused and
// the unfused kernels agree bit for bit): v_exp_f32 + v_rcp_f32 (1 ulp each; the result is rounded to bf16 right after) instead of
// the IEEE division's ten instructions
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
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
