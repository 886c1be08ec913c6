// Attention dropout (reference: F.scaled_dot_product_attention(..., dropout_p=atten_dropout) in training mode,
// src/model/layers/attn.py:122-127; AttentionConfig.atten_dropout defaults to 0.1, attn.py:22).
//
// SDPA semantics: P = softmax(QK^T/sqrt(d)); O = (P .* keep / (1-p)) V with keep ~ Bernoulli(1-p) per (b, h, q, k).
// torch's own mask comes from its Philox stream and is not reproducible outside torch; what the kernels need is a
// counter-based keep(seed, b, h, q, k) that forward, dK/dV and dQ regenerate identically although they hold the
// score tile in different orientations.  It is built like simple tabulation hashing (3-wise independent):
//     W(q, k>>1) = A(seed, b*H+h, q)  xor  B(seed, b*H+h, k>>1)          two well-mixed 32-bit words
//     keep(q, k) = halfword(W, k & 1) >= thr,   thr = round(p * 65536)   (p is realised to 1/65536)
// so a lane that holds one query and runs of consecutive keys (forward, dQ) spends one xor per two elements, and a
// lane that holds one key and runs of queries (dK/dV) one xor per element, both plus compare + select.
// The same functions, restated in numpy, are the oracle's mask (oracle/gaot_oracle.py: dropout_keep_mask).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gdrop {

struct Drop {
    const unsigned long long* seed;   // device pointer: read at kernel start (hipGraph replays see fresh values)
    unsigned thr;                     // keep iff 16-bit uniform >= thr
    float inv_keep;                   // 1 / (1 - thr/65536)
    float keep;                       // 1 - thr/65536
    int head0, heads_total;           // the launch's head 0 is head `head0` of `heads_total` (a rank of a head- / sequence-parallel
                                      // step owns a slice of the heads): the mask is keyed by the GLOBAL head, so a head draws the
                                      // same mask wherever it runs and a sharded step equals the unsharded one with dropout on
    __host__ __device__ __forceinline__ int bh(int b, int head) const { return b * heads_total + head0 + head; }
};

inline Drop make_drop(const unsigned long long* seed, float p, int H, int head0 = 0, int heads_total = 0) {
    long t = (long)(p * 65536.0 + 0.5);
    if (t < 0) t = 0;
    if (t > 65535) t = 65535;
    Drop d;
    d.seed = seed;
    d.thr = (unsigned)t;
    d.keep = 1.0f - (float)t / 65536.0f;
    d.inv_keep = 1.0f / d.keep;
    d.head0 = heads_total > 0 ? head0 : 0;
    d.heads_total = heads_total > 0 ? heads_total : H;
    return d;
}

__host__ __device__ __forceinline__ uint32_t mix_a(uint32_t x) {   // "lowbias32"
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t mix_b(uint32_t x) {   // a second, different finaliser
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}
__host__ __device__ __forceinline__ uint32_t row_key(unsigned long long seed, int bh) {
    return mix_a((uint32_t)seed + 0x9E3779B9u * (uint32_t)(bh + 1));
}
__host__ __device__ __forceinline__ uint32_t col_key(unsigned long long seed, int bh) {
    return mix_b((uint32_t)(seed >> 32) + 0x85EBCA6Bu * (uint32_t)(bh + 1));
}
__host__ __device__ __forceinline__ uint32_t row_word(uint32_t rk, uint32_t q) { return mix_a(rk + q); }
__host__ __device__ __forceinline__ uint32_t col_word(uint32_t ck, uint32_t kpair) { return mix_b(ck + kpair); }
__host__ __device__ __forceinline__ bool keep_elem(uint32_t rk, uint32_t ck, uint32_t q, uint32_t k, uint32_t thr) {
    const uint32_t w = row_word(rk, q) ^ col_word(ck, k >> 1);
    return ((k & 1) ? (w >> 16) : (w & 0xffffu)) >= thr;
}

}  // namespace gdrop
