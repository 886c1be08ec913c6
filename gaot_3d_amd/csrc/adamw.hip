// Fused multi-tensor AdamW step (reference src/trainer/optimizers.py:210 -> torch.optim.AdamW(params, lr, weight_decay),
// defaults betas (0.9, 0.999), eps 1e-8, decoupled weight decay, no amsgrad).  One launch updates up to 48 parameter
// tensors; the arithmetic follows torch's single-tensor reference update term by term:
//   p *= 1 - lr*wd;  m += (g - m)(1 - b1);  v = v*b2 + (1 - b2) g*g;     (1 - b) formed in double on the host, as torch does)
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step counter t and the learning rate live in device memory (t is advanced by the first launch of a step), so a
// captured hipGraph of the whole training step replays correctly and a host LR schedule only rewrites one float.
// HBM bound: 16 B read + 12 B written per parameter.
#include "common.h"

namespace {

constexpr int AW_MAX = 48;
constexpr int AW_CHUNK = 4096;   // elements per workgroup

struct AwTable {
    float* p[AW_MAX];
    const float* g[AW_MAX];
    float* m[AW_MAX];
    float* v[AW_MAX];
    int64_t n[AW_MAX];
    int first_block[AW_MAX + 1];   // prefix sum of ceil(n / AW_CHUNK)
    int count;
};

__global__ void k_adamw_tick(float* step) { *step += 1.0f; }

__global__ __launch_bounds__(256) void k_adamw(AwTable t, const float* __restrict__ lr_p, const float* __restrict__ step_p,
                                               float b1, float b2, float omb1, float omb2, float eps, float wd) {
    // which tensor does this workgroup belong to? (<= 48 entries: linear scan by one wave is fine)
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const int64_t base = (int64_t)((int)blockIdx.x - t.first_block[ti]) * AW_CHUNK;
    const int64_t n = t.n[ti];
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti];
    float* __restrict__ v = t.v[ti];
    const float lr = *lr_p;
    const double step = (double)*step_p;
    const float bc1 = (float)(1.0 - pow((double)b1, step));
    const float bc2s = sqrtf((float)(1.0 - pow((double)b2, step)));
    const float step_size = lr / bc1;
    const float decay = 1.0f - lr * wd;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        pv *= decay;
        mv = mv + (gv - mv) * omb1;
        vv = vv * b2 + omb2 * gv * gv;
        const float denom = sqrtf(vv) / bc2s + eps;
        pv = pv - step_size * (mv / denom);
    };
#pragma unroll
    for (int it = 0; it < AW_CHUNK / (256 * 4); ++it) {
        const int64_t i = base + ((int64_t)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int j = 0; j < 4 && i + j < n; ++j) upd(p[i + j], g[i + j], m[i + j], v[i + j]);
        }
    }
}

}  // namespace

extern "C" int gaot_adamw_step(const gaot_adamw_tensor_t* tensors, int num_tensors, const float* lr, float* step,
                               double beta1, double beta2, double eps, double weight_decay, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_tensors >= 0, "negative tensor count");
    GAOT_CHECK_ARG(lr && step, "lr and step must be device pointers");
    GAOT_CHECK_ARG(num_tensors == 0 || tensors, "null tensor table");
    hipStream_t st = (hipStream_t)stream;
    GAOT_KLAUNCH(k_adamw_tick, dim3(1), dim3(1), 0, st, step);
    for (int t0 = 0; t0 < num_tensors; t0 += AW_MAX) {
        AwTable tb;
        tb.count = 0;
        int blocks = 0;
        for (int i = t0; i < num_tensors && tb.count < AW_MAX; ++i) {
            const gaot_adamw_tensor_t& e = tensors[i];
            GAOT_CHECK_ARG(e.numel >= 0, "negative tensor size");
            if (e.numel == 0) continue;
            GAOT_CHECK_ARG(e.param && e.grad && e.exp_avg && e.exp_avg_sq, "null pointer in tensor table");
            const int c = tb.count++;
            tb.p[c] = e.param; tb.g[c] = e.grad; tb.m[c] = e.exp_avg; tb.v[c] = e.exp_avg_sq; tb.n[c] = e.numel;
            tb.first_block[c] = blocks;
            blocks += (int)ceil_div(e.numel, AW_CHUNK);
        }
        tb.first_block[tb.count] = blocks;
        if (blocks)
            GAOT_KLAUNCH(k_adamw, dim3(blocks), dim3(256), 0, st, tb, lr, (const float*)step, (float)beta1, (float)beta2,
                               (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
