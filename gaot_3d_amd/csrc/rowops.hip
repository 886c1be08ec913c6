// HBM-bound row / element kernels of the latent Transformer and the model glue.  Each stands in
// for a chain of ATen elementwise / reduction ops in the reference:
//   rmsnorm      RMSNorm.forward                      src/model/layers/attn.py:174-178
//   rope         rotate_queries_or_keys (1-D RoPE)    attn.py:118-120 (rotary_embedding_torch)
//   swiglu       silu(w1 x) * w3 x                    attn.py:156
//   act_bwd      autograd of F.gelu / ReLU            mlp.py:330-331, geoembed.py:37
//   patchify     view/permute/contiguous              gaot_3d.py:199-202, 218-220
//   mse          nn.MSELoss                           src/trainer/base.py:56
//   colsum       bias gradients (sum over rows)
// All reductions are two-stage in a fixed order (bit-reproducible), 16-byte vector accesses.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// RMSNorm: one wave per row, lanes stride the row in float4
// ---------------------------------------------------------------------------------------------
__global__ void k_rmsnorm_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                              float* __restrict__ rstd, unsigned short* __restrict__ yb, int64_t rows, int d, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * d);
    float ss = 0.f;
    for (int i = lane; i < d / 4; i += 64) {
        const float4 v = xr[i];
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd) rstd[row] = r;
    float4* yr = reinterpret_cast<float4*>(y + row * d);
    const float4* wr = reinterpret_cast<const float4*>(w);
    for (int i = lane; i < d / 4; i += 64) {
        const float4 v = xr[i], g = wr[i];
        const float4 o = make_float4(v.x * r * g.x, v.y * r * g.y, v.z * r * g.z, v.w * r * g.w);
        yr[i] = o;
        if (yb) {   // the same row rounded to bf16: the A operand of the GEMM that consumes it (q|k|v, w1|w3)
            const __bf16 b0 = (__bf16)o.x, b1 = (__bf16)o.y, b2 = (__bf16)o.z, b3 = (__bf16)o.w;
            uint2 pk;
            pk.x = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
            pk.y = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
            reinterpret_cast<uint2*>(yb + row * d)[i] = pk;
        }
    }
}

// dx = r*w*dy - x*r^3*mean(x*w*dy);  dw partial per block = sum_rows dy*x*r
constexpr int RN_ROWS_PER_WAVE = 8;
__global__ void k_rmsnorm_bwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                              const float* __restrict__ rstd, const float* __restrict__ dx_add, const float* __restrict__ dx_add2,
                              float* __restrict__ dx, float* __restrict__ dw_part, int64_t rows, int d) {
    extern __shared__ float sm[];  // [4][d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = d / 4;
    const float4* wr = reinterpret_cast<const float4*>(w);
    float4 dwacc[4];  // supports d <= 1024
#pragma unroll
    for (int j = 0; j < 4; ++j) dwacc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RN_ROWS_PER_WAVE;
    for (int rr = 0; rr < RN_ROWS_PER_WAVE; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= rows) break;
        const float4* xr = reinterpret_cast<const float4*>(x + row * d);
        const float4* gr = reinterpret_cast<const float4*>(dy + row * d);
        const float r = rstd[row];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const float4 v = xr[i], g = gr[i], ww = wr[i];
                dot += v.x * g.x * ww.x + v.y * g.y * ww.y + v.z * g.z * ww.z + v.w * g.w * ww.w;
            }
        }
        dot = wave_sum(dot);
        const float c = dot * r * r * r / (float)d;
        float4* dxr = reinterpret_cast<float4*>(dx + row * d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const float4 v = xr[i], g = gr[i], ww = wr[i];
                // dx_add: the gradient that reaches x through its other consumer (the block's residual), added here
                float4 e = dx_add ? reinterpret_cast<const float4*>(dx_add + row * d)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (dx_add2) {      // a third consumer of x (the U-ViT skip tap): (dx + dx_add) + dx_add2, the order of two accumulation passes
                    const float4 e2 = reinterpret_cast<const float4*>(dx_add2 + row * d)[i];
                    dxr[i] = make_float4((r * ww.x * g.x - v.x * c + e.x) + e2.x, (r * ww.y * g.y - v.y * c + e.y) + e2.y,
                                         (r * ww.z * g.z - v.z * c + e.z) + e2.z, (r * ww.w * g.w - v.w * c + e.w) + e2.w);
                } else
                dxr[i] = make_float4(r * ww.x * g.x - v.x * c + e.x, r * ww.y * g.y - v.y * c + e.y,
                                     r * ww.z * g.z - v.z * c + e.z, r * ww.w * g.w - v.w * c + e.w);
                dwacc[j].x += g.x * v.x * r;
                dwacc[j].y += g.y * v.y * r;
                dwacc[j].z += g.z * v.z * r;
                dwacc[j].w += g.w * v.w * r;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) reinterpret_cast<float4*>(sm + wave * d)[i] = dwacc[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256)
        dw_part[(int64_t)blockIdx.x * d + i] = sm[i] + sm[d + i] + sm[2 * d + i] + sm[3 * d + i];
}

// d == 256 (the Transformer's width): a lane owns ONE float4 of a row, so the eight rows of a wave are loaded TOGETHER (x, dy and
// the residual's gradient: 24 independent 16-byte loads in flight per lane) before the eight row reductions -- the generic kernel
// above walks its rows one after the other, each a dependent load -> reduce -> store chain (17.3 us at [16 384, 256]: latency
// bound at 3.9 TB/s).  Same arithmetic, same summation order, same partial layout.
__global__ __launch_bounds__(256) void k_rmsnorm_bwd_d256(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, const float* __restrict__ rstd,
                                                          const float* __restrict__ dx_add, const float* __restrict__ dx_add2,
                                                          float* __restrict__ dx, float* __restrict__ dw_part, int64_t rows) {
    constexpr int d = 256, R = RN_ROWS_PER_WAVE;
    __shared__ float sm[4 * d];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 ww = reinterpret_cast<const float4*>(w)[lane];
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    float4 v[R], g[R], e[R];
    float r[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int64_t row = row0 + rr < rows ? row0 + rr : rows - 1;      // clamped: the tail rows are computed and not stored
        v[rr] = reinterpret_cast<const float4*>(x + row * d)[lane];
        g[rr] = reinterpret_cast<const float4*>(dy + row * d)[lane];
        e[rr] = dx_add ? reinterpret_cast<const float4*>(dx_add + row * d)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        r[rr] = rstd[row];
    }
    float dot[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
        dot[rr] = v[rr].x * g[rr].x * ww.x + v[rr].y * g[rr].y * ww.y + v[rr].z * g[rr].z * ww.z + v[rr].w * g[rr].w * ww.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) dot[rr] += __shfl_xor(dot[rr], o, 64);
    float4 dwacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        if (row0 + rr >= rows) break;
        const float rs = r[rr], c = dot[rr] * rs * rs * rs / (float)d;
        float4 o = make_float4(rs * ww.x * g[rr].x - v[rr].x * c + e[rr].x, rs * ww.y * g[rr].y - v[rr].y * c + e[rr].y,
                               rs * ww.z * g[rr].z - v[rr].z * c + e[rr].z, rs * ww.w * g[rr].w - v[rr].w * c + e[rr].w);
        if (dx_add2) {      // a third consumer of x (the U-ViT skip tap), added last: the order of two accumulation passes
            const float4 e2 = reinterpret_cast<const float4*>(dx_add2 + (row0 + rr) * d)[lane];
            o = make_float4(o.x + e2.x, o.y + e2.y, o.z + e2.z, o.w + e2.w);
        }
        reinterpret_cast<float4*>(dx + (row0 + rr) * d)[lane] = o;
        dwacc.x += g[rr].x * v[rr].x * rs;
        dwacc.y += g[rr].y * v[rr].y * rs;
        dwacc.z += g[rr].z * v[rr].z * rs;
        dwacc.w += g[rr].w * v[rr].w * rs;
    }
    reinterpret_cast<float4*>(sm + wave * d)[lane] = dwacc;
    __syncthreads();
    dw_part[(int64_t)blockIdx.x * d + threadIdx.x] = sm[threadIdx.x] + sm[d + threadIdx.x] + sm[2 * d + threadIdx.x] + sm[3 * d + threadIdx.x];
}

// out[n] = sum_{p<parts} part[p][n]; 8 columns x 32 part-lanes per block (the partial tables are a few hundred rows of a
// few hundred columns: many short blocks instead of 8 long serial ones -- 13 -> 4 us), fixed summation order
constexpr int RP_COLS = 8, RP_LANES = 32;
__global__ void k_reduce_parts(const float* __restrict__ part, int64_t parts, int64_t n, float* __restrict__ out) {
    __shared__ float sm[RP_LANES][RP_COLS + 1];
    const int cx = threadIdx.x % RP_COLS, ry = threadIdx.x / RP_COLS;
    const int64_t i = (int64_t)blockIdx.x * RP_COLS + cx;
    float s = 0.f;
    if (i < n)
        for (int64_t p = ry; p < parts; p += RP_LANES) s += part[p * n + i];
    sm[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < RP_LANES; ++j) t += sm[j][cx];
        out[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// column sums of X[M][N] (ld): partial per row-chunk, then k_reduce_parts
// ---------------------------------------------------------------------------------------------
__global__ void k_colsum_part(const float* __restrict__ x, int64_t M, int64_t N, int64_t ld, int64_t rows_per_chunk,
                              float* __restrict__ part) {
    __shared__ float sm[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;  // 32 columns x 8 row lanes
    const int64_t n = (int64_t)blockIdx.x * 32 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < M) ? r0 + rows_per_chunk : M;
    float s = 0.f;
    if (n < N)
        for (int64_t r = r0 + ry; r < r1; r += 8) s += x[r * ld + n];
    sm[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += sm[j][cx];
        part[(int64_t)blockIdx.y * N + n] = t;
    }
}

// the same with 16-byte loads: 32 float4 column groups x 8 row lanes per block (N, ld multiples of 4, 16-byte aligned x)
__global__ void k_colsum_part4(const float* __restrict__ x, int64_t M, int64_t N, int64_t ld, int64_t rows_per_chunk,
                               float* __restrict__ part) {
    __shared__ float4 sm[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int64_t n = ((int64_t)blockIdx.x * 32 + cx) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < M) ? r0 + rows_per_chunk : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        // four rows requested together, added in row order (the sums are the one-row loop's; 8 M-row bias gradients: 279 us at 3.7 TB/s)
        int64_t r = r0 + ry;
        for (; r + 24 < r1; r += 32) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (r + 8 * u) * ld + n);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; r < r1; r += 8) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ld + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { t.x += sm[j][cx].x; t.y += sm[j][cx].y; t.z += sm[j][cx].z; t.w += sm[j][cx].w; }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + n) = t;
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE, in place: rows of width ld; nheads heads of head_dim = 2 * HP starting at column col0.  pos = row % S.
// ---------------------------------------------------------------------------------------------
__global__ void k_rope(float* __restrict__ x, int64_t rows, int64_t ld, int col0, int nheads, int HP, int S,
                       const float* __restrict__ freqs, float sign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = rows * nheads * HP;
    if (i >= n) return;
    const int p = (int)(i % HP);
    const int hd = (int)((i / HP) % nheads);
    const int64_t row = i / ((int64_t)HP * nheads);
    const float ang = (float)(row % S) * freqs[p];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    sn *= sign;
    float2* px = reinterpret_cast<float2*>(x + row * ld + col0 + hd * (2 * HP) + 2 * p);
    const float2 v = *px;
    *px = make_float2(v.x * cs - v.y * sn, v.y * cs + v.x * sn);
}

// ---------------------------------------------------------------------------------------------
// SwiGLU on a fused [rows][2F] buffer (a = cols 0..F-1 = w1 x, g = cols F..2F-1 = w3 x)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ void k_swiglu_fwd(const float* __restrict__ ag, float* __restrict__ u, int64_t rows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    const int fv = F / 4;
    if (i >= rows * fv) return;
    const int64_t r = i / fv;
    const int c = (int)(i % fv);
    const float4 a = reinterpret_cast<const float4*>(ag + r * 2 * F)[c];
    const float4 g = reinterpret_cast<const float4*>(ag + r * 2 * F + F)[c];
    reinterpret_cast<float4*>(u + r * F)[c] =
        make_float4(a.x * sigmoid_f(a.x) * g.x, a.y * sigmoid_f(a.y) * g.y, a.z * sigmoid_f(a.z) * g.z,
                    a.w * sigmoid_f(a.w) * g.w);
}

__global__ void k_swiglu_bwd(const float* __restrict__ ag, const float* __restrict__ du, float* __restrict__ dag,
                             int64_t rows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int fv = F / 4;
    if (i >= rows * fv) return;
    const int64_t r = i / fv;
    const int c = (int)(i % fv);
    const float4 a = reinterpret_cast<const float4*>(ag + r * 2 * F)[c];
    const float4 g = reinterpret_cast<const float4*>(ag + r * 2 * F + F)[c];
    const float4 d = reinterpret_cast<const float4*>(du + r * F)[c];
    auto f = [](float av, float gv, float dv, float& da, float& dg) {
        const float s = sigmoid_f(av);
        da = dv * gv * s * (1.f + av * (1.f - s));
        dg = dv * av * s;
    };
    float4 da, dg;
    f(a.x, g.x, d.x, da.x, dg.x);
    f(a.y, g.y, d.y, da.y, dg.y);
    f(a.z, g.z, d.z, da.z, dg.z);
    f(a.w, g.w, d.w, da.w, dg.w);
    reinterpret_cast<float4*>(dag + r * 2 * F)[c] = da;
    reinterpret_cast<float4*>(dag + r * 2 * F + F)[c] = dg;
}

// bf16-in-memory variants (the FFN intermediates of the bf16 path): 8 elements per thread, fp32 arithmetic
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f[2 * i]) |
               ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)f[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ void k_cast_bf16(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n8, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n8) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        reinterpret_cast<uint4*>(dst)[i] = pack8(f);
    } else if (i == n8) {
        for (int64_t j = 8 * n8; j < n; ++j) dst[j] = __builtin_bit_cast(unsigned short, (__bf16)src[j]);
    }
}

// the same for up to 64 tensors in one launch (the Transformer's weights, once per forward): a workgroup finds its
// tensor in the table of first-block prefix sums
constexpr int CM_MAX = 64;
struct CastTable {
    const float* src[CM_MAX];
    unsigned short* dst[CM_MAX];
    int64_t n[CM_MAX];
    int first_block[CM_MAX + 1];
    int count;
};
__global__ void k_cast_bf16_multi(CastTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const int64_t i = (int64_t)((int)blockIdx.x - t.first_block[ti]) * blockDim.x + threadIdx.x;
    const int64_t n = t.n[ti], n8 = n / 8;
    const float* src = t.src[ti];
    unsigned short* dst = t.dst[ti];
    if (i < n8) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        reinterpret_cast<uint4*>(dst)[i] = pack8(f);
    } else if (i == n8) {
        for (int64_t j = 8 * n8; j < n; ++j) dst[j] = __builtin_bit_cast(unsigned short, (__bf16)src[j]);
    }
}

__global__ void k_swiglu_fwd_bf16(const unsigned short* __restrict__ ag, unsigned short* __restrict__ u, int64_t rows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte chunk index
    const int fv = F / 8;
    if (i >= rows * fv) return;
    const int64_t r = i / fv;
    const int c = (int)(i % fv);
    float a[8], g[8], o[8];
    unpack8(reinterpret_cast<const uint4*>(ag + r * 2 * F)[c], a);
    unpack8(reinterpret_cast<const uint4*>(ag + r * 2 * F + F)[c], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = a[j] * sigmoid_fast(a[j]) * g[j];
    reinterpret_cast<uint4*>(u + r * F)[c] = pack8(o);
}

__global__ void k_swiglu_bwd_bf16(const unsigned short* __restrict__ ag, const unsigned short* __restrict__ du,
                                  unsigned short* __restrict__ dag, int64_t rows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int fv = F / 8;
    if (i >= rows * fv) return;
    const int64_t r = i / fv;
    const int c = (int)(i % fv);
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8(reinterpret_cast<const uint4*>(ag + r * 2 * F)[c], a);
    unpack8(reinterpret_cast<const uint4*>(ag + r * 2 * F + F)[c], g);
    unpack8(reinterpret_cast<const uint4*>(du + r * F)[c], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float s = sigmoid_fast(a[j]);
        da[j] = d[j] * g[j] * s * (1.f + a[j] * (1.f - s));
        dg[j] = d[j] * a[j] * s;
    }
    reinterpret_cast<uint4*>(dag + r * 2 * F)[c] = pack8(da);
    reinterpret_cast<uint4*>(dag + r * 2 * F + F)[c] = pack8(dg);
}

// h = act(z) for the activations the GEMM epilogue does not carry (GAOT_ACT_* >= 4)
__global__ void k_act_fwd(const float* __restrict__ z, float* __restrict__ h, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    h[i] = gaot_act_fwd(z[i], act);
}

// dz = dh * act'(z)
__global__ void k_act_bwd(const float* __restrict__ z, const float* __restrict__ dh, float* __restrict__ dz, int64_t n,
                          int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    dz[i] = dh[i] * gaot_act_grad(v, act);
}

// out = a + alpha * b (b may be broadcast over rows with period `period` elements; period == n: plain)
__global__ void k_axpy(const float* __restrict__ a, const float* __restrict__ b, float alpha, float* __restrict__ out,
                       int64_t n, int64_t period) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] + alpha * b[i % period];
}

// ---------------------------------------------------------------------------------------------
// patchify: tokens[b][(pd,ph,pw)][(i,j,k,c)] <-> grid[b][(pd*P+i, ph*P+j, pw*P+k)][c]; C % 4 == 0
// ---------------------------------------------------------------------------------------------
__global__ void k_patchify(const float* __restrict__ src, float* __restrict__ dst, int B, int Dd, int Hh, int Ww, int P,
                           int C, int to_tokens) {
    const int cv = C / 4;
    const int64_t total = (int64_t)B * Dd * Hh * Ww * cv;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cv);
    int64_t t = i / cv;                 // grid-order node index within the batch of grids
    const int w = (int)(t % Ww); t /= Ww;
    const int h = (int)(t % Hh); t /= Hh;
    const int d = (int)(t % Dd);
    const int b = (int)(t / Dd);
    const int nH = Hh / P, nW = Ww / P, nD = Dd / P;
    const int64_t tok = ((int64_t)(d / P) * nH + (h / P)) * nW + (w / P);
    const int64_t within = ((int64_t)(d % P) * P + (h % P)) * P + (w % P);
    const int64_t gidx = (((int64_t)b * Dd + d) * Hh + h) * Ww + w;
    const int64_t tidx = (((int64_t)b * nD * nH * nW + tok) * P * P * P + within);
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    if (to_tokens) d4[tidx * cv + c] = s4[gidx * cv + c];
    else d4[gidx * cv + c] = s4[tidx * cv + c];
}

// ---------------------------------------------------------------------------------------------
// MSE: partial sums in double, fixed-order final; backward reads the upstream scalar from memory
// ---------------------------------------------------------------------------------------------
__global__ void k_mse_part(const float* __restrict__ p, const float* __restrict__ t, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)p[i] - (double)t[i];
        s += d * d;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one wave: lane l sums parts l, l+64, ... in order, then a fixed butterfly (was one thread walking all parts: 60 us)
__global__ void k_mse_final(const double* __restrict__ part, int nparts, double inv_n, float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) *loss = (float)(s * inv_n);
}
__global__ void k_mse_bwd(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                          const float* __restrict__ gscalar, float coef, float* __restrict__ dp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dp[i] = (p[i] - t[i]) * coef * (*gscalar);
}


// ---------------------------------------------------------------------------------------------
// multi-scale mix: out[n][c] = sum_s softmax(logits[n][:])_s * x_s[n][c]   (reference magno.py:590-594)
// 32 lanes per row (C == 32), two rows per wave
// ---------------------------------------------------------------------------------------------
constexpr int MAX_SCALES = 8;
struct ScalePtrs { const float* x[MAX_SCALES]; float* dx[MAX_SCALES]; };

__global__ void k_scale_mix_fwd(ScalePtrs p, int ns, const float* __restrict__ logits, float* __restrict__ out,
                                float* __restrict__ wsave, int64_t n) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int c = threadIdx.x & 31;
    if (row >= n) return;
    float mx = -INFINITY;
    for (int s = 0; s < ns; ++s) mx = fmaxf(mx, logits[row * ns + s]);
    float den = 0.f;
    for (int s = 0; s < ns; ++s) den += expf(logits[row * ns + s] - mx);
    float acc = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float w = expf(logits[row * ns + s] - mx) / den;
        if (c == 0) wsave[row * ns + s] = w;
        acc += w * p.x[s][row * 32 + c];
    }
    out[row * 32 + c] = acc;
}

__global__ void k_scale_mix_bwd(ScalePtrs p, int ns, const float* __restrict__ w, const float* __restrict__ dout,
                                float* __restrict__ dlogits, int64_t n) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int c = threadIdx.x & 31;
    if (row >= n) return;
    const float g = dout[row * 32 + c];
    float gs[MAX_SCALES];
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_SCALES; ++s) {
        gs[s] = 0.f;
        if (s < ns) {
            float v = g * p.x[s][row * 32 + c];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            gs[s] = v;
            const float ws = w[row * ns + s];
            tot += ws * v;
            p.dx[s][row * 32 + c] = ws * g;
        }
    }
    if (c == 0) {
#pragma unroll
        for (int s = 0; s < MAX_SCALES; ++s)
            if (s < ns) dlogits[row * ns + s] = w[row * ns + s] * (gs[s] - tot);
    }
}

unsigned blocks_for(int64_t n, int tb = 256) { return (unsigned)std::max<int64_t>(1, ceil_div(n, tb)); }

}  // namespace

extern "C" int gaot_rmsnorm_fwd(const float* x, const float* weight, float* y, float* rstd, void* y_bf16, int64_t rows,
                                int dim, float eps, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && dim > 0 && dim % 4 == 0, "dim must be a positive multiple of 4");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(x && weight && y, "null pointer");
    GAOT_KLAUNCH(k_rmsnorm_fwd, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, weight, y,
                       rstd, (unsigned short*)y_bf16, rows, dim, eps);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" size_t gaot_rmsnorm_bwd_workspace_bytes(int64_t rows, int dim) {
    return sizeof(float) * (size_t)(ceil_div(rows, 4 * RN_ROWS_PER_WAVE) * dim) + 64;
}

static int rmsnorm_bwd_impl(const float* x, const float* weight, const float* dy, const float* rstd, const float* dx_add,
                            const float* dx_add2, float* dx, float* dweight, int64_t rows, int dim, void* workspace, size_t workspace_bytes,
                            gaot_stream_t stream) {
    GAOT_CHECK_ARG(rows >= 0 && dim > 0 && dim % 4 == 0 && dim <= 1024, "dim must be a multiple of 4, <= 1024");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_rmsnorm_bwd_workspace_bytes(rows, dim), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        if (dweight) hipMemsetAsync(dweight, 0, sizeof(float) * dim, st);
        return GAOT_OK;
    }
    GAOT_CHECK_ARG(x && weight && dy && rstd && dx && workspace, "null pointer");
    const int64_t nblk = ceil_div(rows, 4 * RN_ROWS_PER_WAVE);
    float* part = (float*)workspace;
    if (dim == 256 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)dx_add2 | (uintptr_t)weight) & 15) == 0)
        GAOT_KLAUNCH(k_rmsnorm_bwd_d256, dim3((unsigned)nblk), dim3(256), 0, st, x, weight, dy, rstd, dx_add, dx_add2, dx, part, rows);
    else
        GAOT_KLAUNCH(k_rmsnorm_bwd, dim3((unsigned)nblk), dim3(256), sizeof(float) * 4 * dim, st, x, weight, dy, rstd,
                           dx_add, dx_add2, dx, part, rows, dim);
    // dweight == NULL: the gaot_rmsnorm_bwd_parts(rows) partial rows stay in the workspace for gaot_reduce_multi (32 lanes)
    if (dweight) GAOT_KLAUNCH(k_reduce_parts, dim3(blocks_for(dim, RP_COLS)), dim3(256), 0, st, part, nblk, (int64_t)dim, dweight);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_rmsnorm_bwd(const float* x, const float* weight, const float* dy, const float* rstd,
                                const float* dx_add, float* dx, float* dweight, int64_t rows, int dim, void* workspace,
                                size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    return rmsnorm_bwd_impl(x, weight, dy, rstd, dx_add, nullptr, dx, dweight, rows, dim, workspace, workspace_bytes, stream);
}
// the same with a SECOND gradient that reaches x through a third consumer (the long-range skip of the U-ViT, reference attn.py:282-288:
// an encoder block's output feeds the next block AND the mirrored decoder block): dx = (dx + dx_add) + dx_add2 in the same pass --
// stands in for the accumulation pass the autograd engine would run on the two gradients
extern "C" int gaot_rmsnorm_bwd2(const float* x, const float* weight, const float* dy, const float* rstd, const float* dx_add,
                                 const float* dx_add2, float* dx, float* dweight, int64_t rows, int dim, void* workspace,
                                 size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    return rmsnorm_bwd_impl(x, weight, dy, rstd, dx_add, dx_add2, dx, dweight, rows, dim, workspace, workspace_bytes, stream);
}
extern "C" int64_t gaot_rmsnorm_bwd_parts(int64_t rows) { return ceil_div(rows, 4 * RN_ROWS_PER_WAVE); }

extern "C" size_t gaot_colsum_workspace_bytes(int64_t M, int64_t N) {
    const int64_t chunks = std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(M, 64)));
    return sizeof(float) * (size_t)(chunks * N) + 64;
}

extern "C" int gaot_colsum(const float* x, int64_t M, int64_t N, int64_t ld, float* out, void* workspace,
                           size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(M >= 0 && N >= 0, "negative size");
    if (N == 0) return GAOT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        GAOT_CHECK_ARG(out, "null pointer");
        hipMemsetAsync(out, 0, sizeof(float) * N, st);
        return GAOT_OK;
    }
    GAOT_CHECK_ARG(x && workspace && workspace_bytes >= gaot_colsum_workspace_bytes(M, N), "workspace too small");
    const int64_t chunks = std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(M, 64)));
    const int64_t rpc = ceil_div(M, chunks);
    float* part = (float*)workspace;
    if (N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0)
        GAOT_KLAUNCH(k_colsum_part4, dim3((unsigned)ceil_div(N, 128), (unsigned)chunks), dim3(256), 0, st, x, M, N, ld,
                           rpc, part);
    else
        GAOT_KLAUNCH(k_colsum_part, dim3((unsigned)ceil_div(N, 32), (unsigned)chunks), dim3(256), 0, st, x, M, N, ld,
                           rpc, part);
    // out == NULL: the gaot_colsum_parts(M) partial rows stay in the workspace for gaot_reduce_multi (32 lanes)
    if (out) GAOT_KLAUNCH(k_reduce_parts, dim3(blocks_for(N, RP_COLS)), dim3(256), 0, st, part, chunks, N, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
extern "C" int64_t gaot_colsum_parts(int64_t M) { return std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(M, 64))); }

// ---------------------------------------------------------------------------------------------
// gaot_reduce_multi: out_j[i] = sum_{p < parts_j} part_j[p][i] for up to 64 (partials, output) pairs in ONE launch -- the
// completion of every split-K weight gradient, RMSNorm weight gradient and bias column sum of a backward pass, deferred to
// its end (nothing reads them before the optimizer step): ~90 5-us launches of a configs[1] step become two.  A pair keeps the
// summation order of the pass it replaces: `lanes` part-lanes each sum every lanes-th partial, then the lane sums are added in
// lane order (4: k_splitk_reduce<64>, 16: k_splitk_reduce<16>, 32: k_reduce_parts) -- bit-identical results, no atomics.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int RM_MAX = 64;
struct ReduceTable {
    const float* part[RM_MAX];
    float* out[RM_MAX];
    long long n[RM_MAX];
    int parts[RM_MAX];
    int lanes[RM_MAX];     // 4, 16 or 32
    int vec[RM_MAX];       // outputs per thread: 4 (n % 4 == 0, 16-byte aligned) or 1
    int wide[RM_MAX];      // lanes == 4, 16-byte columns, n >= 16 384: one thread per column, lane sums in registers
    int blk0[RM_MAX + 1];  // first block of pair j
    int count;
};
__global__ __launch_bounds__(256) void k_reduce_multi(const ReduceTable t) {
    __shared__ float4 red[256];
    int j = 0;
    while (j + 1 < t.count && (int)blockIdx.x >= t.blk0[j + 1]) ++j;     // uniform: <= 64 scalar compares
    const int lanes = t.lanes[j], cols = 256 / lanes, vec = t.vec[j];
    const int o = threadIdx.x % cols, sl = threadIdx.x / cols;
    const long long n = t.n[j];
    const long long i = ((long long)(blockIdx.x - t.blk0[j]) * cols + o) * vec;
    const float* __restrict__ part = t.part[j];
    const int parts = t.parts[j];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t.wide[j]) {
        // wide tables of a few partials (the Transformer's split-K weight gradients: n = 65 K .. 524 K outputs, 8 .. 32 partials):
        // a thread owns one 16-byte column and keeps the FOUR lane sums itself -- no LDS, no barrier, every load independent and
        // non-temporal (the partials were written once, long ago, and are read once); ((l0 + l1) + l2) + l3 as above
        typedef float f4v __attribute__((ext_vector_type(4)));
        const long long c = ((long long)(blockIdx.x - t.blk0[j]) * 256 + threadIdx.x) * 4;
        if (c >= n) return;
        f4v acc[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[l] = (f4v){0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ p0 = part + c;
        int s = 0;
        for (; s + 4 <= parts; s += 4) {
            f4v a[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) a[l] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p0 + (long long)(s + l) * n));
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[l] += a[l];
        }
        for (int l = 0; s < parts; ++s, ++l) acc[l] += __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p0 + (long long)s * n));
        const f4v r = ((acc[0] + acc[1]) + acc[2]) + acc[3];
        *reinterpret_cast<f4v*>(t.out[j] + c) = r;
        return;
    }
    if (i < n) {
        if (vec == 4) {
            for (int s = sl; s < parts; s += lanes) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v a = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(part + (long long)s * n + i));
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
        } else {
            for (int s = sl; s < parts; s += lanes) v.x += part[(long long)s * n + i];
        }
    }
    red[sl * cols + o] = v;
    __syncthreads();
    if (sl != 0 || i >= n) return;
    v = red[o];
    for (int l = 1; l < lanes; ++l) {
        const float4 a = red[l * cols + o];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (vec == 4) *reinterpret_cast<float4*>(t.out[j] + i) = v;
    else t.out[j][i] = v.x;
}
}  // namespace

extern "C" int gaot_reduce_multi(const gaot_reduce_desc_t* descs, int count, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(count >= 0 && (count == 0 || descs), "bad descriptor list");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += RM_MAX) {
        ReduceTable t;
        t.count = std::min(RM_MAX, count - base);
        long long blocks = 0;
        for (int j = 0; j < t.count; ++j) {
            const gaot_reduce_desc_t& d = descs[base + j];
            GAOT_CHECK_ARG(d.part && d.out && d.n >= 0 && d.parts >= 1 && (d.lanes == 4 || d.lanes == 16 || d.lanes == 32),
                           "reduce descriptor: null pointer, parts < 1 or lanes not in {4, 16, 32}");
            t.part[j] = d.part; t.out[j] = d.out; t.n[j] = d.n; t.parts[j] = d.parts; t.lanes[j] = d.lanes;
            t.vec[j] = (d.n % 4 == 0 && (((uintptr_t)d.part | (uintptr_t)d.out) & 15) == 0) ? 4 : 1;
            t.wide[j] = (d.lanes == 4 && t.vec[j] == 4 && d.n >= 16384) ? 1 : 0;
            t.blk0[j] = (int)blocks;
            blocks += t.wide[j] ? ceil_div(d.n, 1024LL) : ceil_div(d.n, (long long)(256 / d.lanes) * t.vec[j]);
            GAOT_CHECK_ARG(blocks < 0x7fffffff, "reduce descriptors: too many outputs for one launch");
        }
        t.blk0[t.count] = (int)blocks;
        if (blocks > 0) GAOT_KLAUNCH(k_reduce_multi, dim3((unsigned)blocks), dim3(256), 0, st, t);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_rope(float* x, int64_t rows, int64_t ld, int col0, int nheads, int head_dim, int seq_len,
                         const float* freqs, int inverse, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(head_dim >= 2 && head_dim % 2 == 0, "head_dim must be even");
    GAOT_CHECK_ARG(rows >= 0 && nheads > 0 && seq_len > 0 && ld % 2 == 0 && col0 % 2 == 0, "bad shape");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(x && freqs, "null pointer");
    const int64_t n = rows * nheads * (head_dim / 2);
    GAOT_KLAUNCH(k_rope, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, rows, ld, col0, nheads, head_dim / 2,
                       seq_len, freqs, inverse ? -1.f : 1.f);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_swiglu_fwd(const float* ag, float* u, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && F > 0 && F % 4 == 0, "F must be a positive multiple of 4");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(ag && u, "null pointer");
    GAOT_KLAUNCH(k_swiglu_fwd, dim3(blocks_for(rows * (F / 4))), dim3(256), 0, (hipStream_t)stream, ag, u, rows, F);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_swiglu_bwd(const float* ag, const float* du, float* dag, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && F > 0 && F % 4 == 0, "F must be a positive multiple of 4");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(ag && du && dag, "null pointer");
    GAOT_KLAUNCH(k_swiglu_bwd, dim3(blocks_for(rows * (F / 4))), dim3(256), 0, (hipStream_t)stream, ag, du, dag,
                       rows, F);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_cast_bf16(const float* src, void* dst, int64_t n, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0, "negative size");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(src && dst, "null pointer");
    GAOT_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "buffers must be 16-byte aligned");
    const int64_t n8 = n / 8;
    GAOT_KLAUNCH(k_cast_bf16, dim3(blocks_for(n8 + 1)), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n8, n);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// ---- head exchange layouts of the sequence-parallel step (gaot_3d_amd/sharding.py) ------------------------------------
// "rows" layout [rows][ld]: column segments s = 0..nseg-1, each G consecutive groups of w columns starting at col0 (group j
// belongs to rank j: the q, k and v heads of a fused projection; or all heads of an attention output);
// "blocks" layout [G][rows][lw]: block j holds, for every row, the groups of rank j side by side (segment s at poff) -- the
// send / receive buffer of all_to_all_single.  One thread moves 4 consecutive elements; fp32 / bf16 on either side.
namespace {
struct PackSegs { int col0[3], w[3], poff[3], nseg; };
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<unsigned short>(const unsigned short* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(unsigned short* p, float4 v) {
    auto b = [](float f) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); };
    *reinterpret_cast<uint2*>(p) = make_uint2(b(v.x) | (b(v.y) << 16), b(v.z) | (b(v.w) << 16));
}
template <typename TR, typename TB, bool TO_BLOCKS>
__global__ void k_pack_heads(TR* __restrict__ rowsbuf, TB* __restrict__ blocks, int64_t rows, int ld, int lw, int G, PackSegs sg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (row, 4-column group of the packed width G * lw)
    const int per_row = G * lw / 4;
    if (i >= rows * per_row) return;
    const int64_t row = i / per_row;
    const int t = (int)(i % per_row), j = t / (lw / 4), c = (t % (lw / 4)) * 4;     // block j, column c of its row
    int s = 0;
    if (sg.nseg > 1 && c >= sg.poff[1]) s = 1;
    if (sg.nseg > 2 && c >= sg.poff[2]) s = 2;
    const int col = sg.col0[s] + j * sg.w[s] + (c - sg.poff[s]);
    TR* rp = rowsbuf + row * ld + col;
    TB* bp = blocks + ((int64_t)j * rows + row) * lw + c;
    if constexpr (TO_BLOCKS) st4(bp, ld4(rp)); else st4(rp, ld4(bp));
}
}  // namespace

// rows_dtype / blocks_dtype: 0 fp32, 1 bf16.  to_blocks 1: rows layout -> blocks; 0: blocks -> rows layout.
extern "C" int gaot_pack_heads(void* rows_buf, void* blocks_buf, int64_t rows, int ld, int world, int nseg, const int* col0,
                               const int* width, int rows_dtype, int blocks_dtype, int to_blocks, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && world >= 1 && nseg >= 1 && nseg <= 3 && col0 && width, "bad argument");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(rows_buf && blocks_buf, "null pointer");
    PackSegs sg{};
    sg.nseg = nseg;
    int lw = 0;
    for (int s = 0; s < nseg; ++s) {
        GAOT_CHECK_ARG(width[s] > 0 && width[s] % 4 == 0 && col0[s] % 4 == 0 && col0[s] + world * width[s] <= ld, "segments: widths / offsets must be multiples of 4 inside the row");
        sg.col0[s] = col0[s]; sg.w[s] = width[s]; sg.poff[s] = lw;
        lw += width[s];
    }
    GAOT_CHECK_ARG(ld % 4 == 0 && (((uintptr_t)rows_buf | (uintptr_t)blocks_buf) & 15) == 0, "16-byte aligned buffers, ld % 4 == 0");
    const int64_t n = rows * (world * lw / 4);
    const dim3 grid((unsigned)ceil_div(n, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    typedef unsigned short b16;
#define GAOT_PACK(TR, TB)                                                                                                        \
    do {                                                                                                                         \
        if (to_blocks) GAOT_KLAUNCH((k_pack_heads<const TR, TB, true>), grid, blk, 0, st, (const TR*)rows_buf, (TB*)blocks_buf, rows, ld, lw, world, sg); \
        else GAOT_KLAUNCH((k_pack_heads<TR, const TB, false>), grid, blk, 0, st, (TR*)rows_buf, (const TB*)blocks_buf, rows, ld, lw, world, sg);          \
    } while (0)
    if (rows_dtype == 0 && blocks_dtype == 0) GAOT_PACK(float, float);
    else if (rows_dtype == 0 && blocks_dtype == 1) GAOT_PACK(float, b16);
    else if (rows_dtype == 1 && blocks_dtype == 0) GAOT_PACK(b16, float);
    else GAOT_PACK(b16, b16);
#undef GAOT_PACK
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_cast_bf16_multi(const gaot_cast_tensor_t* tensors, int num_tensors, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_tensors >= 0, "negative tensor count");
    GAOT_CHECK_ARG(num_tensors == 0 || tensors, "null tensor table");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < num_tensors; base += CM_MAX) {
        CastTable tb;
        tb.count = 0;
        int blocks = 0;
        for (int i = base; i < num_tensors && i < base + CM_MAX; ++i) {
            const gaot_cast_tensor_t& e = tensors[i];
            GAOT_CHECK_ARG(e.numel >= 0, "negative tensor size");
            if (e.numel == 0) continue;
            GAOT_CHECK_ARG(e.src && e.dst, "null pointer in tensor table");
            GAOT_CHECK_ARG((((uintptr_t)e.src | (uintptr_t)e.dst) & 15) == 0, "buffers must be 16-byte aligned");
            const int c = tb.count++;
            tb.src[c] = e.src; tb.dst[c] = (unsigned short*)e.dst; tb.n[c] = e.numel;
            tb.first_block[c] = blocks;
            blocks += (int)ceil_div(e.numel / 8 + 1, 256);
        }
        tb.first_block[tb.count] = blocks;
        if (tb.count) GAOT_KLAUNCH(k_cast_bf16_multi, dim3(blocks), dim3(256), 0, st, tb);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// fp32 [rows][cols] -> bf16 [cols][rows] for many matrices in ONE launch (32 x 32 tiles through LDS)
struct TransTable {
    const float* src[CM_MAX];
    unsigned short* dst[CM_MAX];
    int rows[CM_MAX], cols[CM_MAX];
    int first_block[CM_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void k_cast_bf16_transpose_multi(TransTable t) {
    __shared__ float tile[32][33];
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const int b = (int)blockIdx.x - t.first_block[ti];
    const int R = t.rows[ti], Cc = t.cols[ti], tc = (Cc + 31) / 32;
    const int r0 = (b / tc) * 32, c0 = (b % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* src = t.src[ti];
    unsigned short* dst = t.dst[ti];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        tile[ty + 8 * j][tx] = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (c < Cc && r < R) dst[(int64_t)c * R + r] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][ty + 8 * j]);
    }
}

extern "C" int gaot_cast_bf16_transpose_multi(const gaot_cast_tensor_t* tensors, const int* rows, const int* cols, int num_tensors,
                                              gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_tensors >= 0, "negative tensor count");
    GAOT_CHECK_ARG(num_tensors == 0 || (tensors && rows && cols), "null tensor table");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < num_tensors; base += CM_MAX) {
        TransTable tb;
        tb.count = 0;
        int blocks = 0;
        for (int i = base; i < num_tensors && i < base + CM_MAX; ++i) {
            const gaot_cast_tensor_t& e = tensors[i];
            GAOT_CHECK_ARG(rows[i] >= 0 && cols[i] >= 0 && e.numel == (int64_t)rows[i] * cols[i], "rows * cols must equal numel");
            if (e.numel == 0) continue;
            GAOT_CHECK_ARG(e.src && e.dst, "null pointer in tensor table");
            const int c = tb.count++;
            tb.src[c] = e.src; tb.dst[c] = (unsigned short*)e.dst; tb.rows[c] = rows[i]; tb.cols[c] = cols[i];
            tb.first_block[c] = blocks;
            blocks += (int)(ceil_div(rows[i], 32) * ceil_div(cols[i], 32));
        }
        tb.first_block[tb.count] = blocks;
        if (tb.count) GAOT_KLAUNCH(k_cast_bf16_transpose_multi, dim3(blocks), dim3(256), 0, st, tb);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_swiglu_fwd_bf16(const void* ag, void* u, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && F > 0 && F % 8 == 0, "F must be a positive multiple of 8");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(ag && u, "null pointer");
    GAOT_CHECK_ARG((((uintptr_t)ag | (uintptr_t)u) & 15) == 0, "buffers must be 16-byte aligned");
    GAOT_KLAUNCH(k_swiglu_fwd_bf16, dim3(blocks_for(rows * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)ag, (unsigned short*)u, rows, F);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_swiglu_bwd_bf16(const void* ag, const void* du, void* dag, int64_t rows, int F, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && F > 0 && F % 8 == 0, "F must be a positive multiple of 8");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(ag && du && dag, "null pointer");
    GAOT_CHECK_ARG((((uintptr_t)ag | (uintptr_t)du | (uintptr_t)dag) & 15) == 0, "buffers must be 16-byte aligned");
    GAOT_KLAUNCH(k_swiglu_bwd_bf16, dim3(blocks_for(rows * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)ag, (const unsigned short*)du, (unsigned short*)dag, rows, F);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_act_fwd(const float* z, float* h, int64_t n, int act, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && act >= 0 && act < GAOT_ACT_COUNT, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(z && h, "null pointer");
    GAOT_KLAUNCH(k_act_fwd, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, h, n, act);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_act_bwd(const float* z, const float* dh, float* dz, int64_t n, int act, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && act >= 0 && act < GAOT_ACT_COUNT, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(z && dh && dz, "null pointer");
    GAOT_KLAUNCH(k_act_bwd, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, dh, dz, n, act);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_axpy(const float* a, const float* b, float alpha, float* out, int64_t n, int64_t period,
                         gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n >= 0 && period > 0, "bad argument");
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(a && b && out, "null pointer");
    GAOT_KLAUNCH(k_axpy, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, out, n, period);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// float4 streaming copy: the measured HBM ceiling the bench line quotes next to the 8 TB/s spec figure.
// VARIANT 0: grid-stride, four plain 16-byte loads in flight per thread (round 4: 4.5-4.9 TB/s).  The others move a CONTIGUOUS chunk per
// workgroup with UNR loads in flight per thread: 1 = non-temporal loads and stores, 2 = plain loads + non-temporal stores, 3 = plain both
// (the copy is read-once / write-once: non-temporal accesses keep it out of the way of the L2 / Infinity-Cache replacement)
typedef float copy_f4 __attribute__((ext_vector_type(4)));
template <int VARIANT, int UNR>
__global__ __launch_bounds__(256) void k_stream_copy(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, int64_t n4, int64_t chunk4) {
    if constexpr (VARIANT == 0) {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const copy_f4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
            dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
        }
        for (; i < n4; i += stride) dst[i] = src[i];
    } else {
        const int64_t lo = (int64_t)blockIdx.x * chunk4, hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
        int64_t i = lo + threadIdx.x;
        for (; i + (UNR - 1) * 256 < hi; i += UNR * 256) {
            copy_f4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if constexpr (VARIANT == 1) v[u] = __builtin_nontemporal_load(src + i + u * 256);
                else v[u] = src[i + u * 256];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if constexpr (VARIANT == 3) dst[i + u * 256] = v[u];
                else __builtin_nontemporal_store(v[u], dst + i + u * 256);
            }
        }
        for (; i < hi; i += 256) dst[i] = src[i];
    }
}

static int stream_copy_variant = 9;    // the fastest measured on MI355X: 6.05-6.08 TB/s at 1 GiB (profiles/archive/r5_b_stream_copy_lab.txt)

extern "C" int gaot_stream_copy_ex(const void* src, void* dst, int64_t bytes, int variant, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(bytes >= 0 && bytes % 16 == 0, "bytes must be a multiple of 16");
    GAOT_CHECK_ARG(variant >= 0 && variant <= 9, "variant 0..9");
    if (bytes == 0) return GAOT_OK;
    GAOT_CHECK_ARG(src && dst && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "16-byte aligned device pointers");
    const int64_t n4 = bytes / 16;
    const copy_f4* s4 = (const copy_f4*)src;
    copy_f4* d4 = (copy_f4*)dst;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) {
        const int64_t blocks = (n4 + 255) / 256;
        GAOT_KLAUNCH((k_stream_copy<0, 4>), dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, s4, d4, n4, (int64_t)0);
    } else {
        // chunks of 64 KiB (4096 float4 = 16 per thread); variants 4-6: 256 KiB chunks with non-temporal / mixed / plain accesses
        // 7: 16 loads in flight, 128-KiB chunks; 8: 8 in flight, 1-MiB chunks; 9: 4 in flight, 16-KiB chunks (all non-temporal)
        const int64_t chunk4 = variant == 7 ? 8192 : variant == 8 ? 65536 : variant == 9 ? 1024 : (variant <= 3 ? 4096 : 16384);
        const int64_t blocks = (n4 + chunk4 - 1) / chunk4;
        GAOT_CHECK_ARG(blocks < (int64_t)1 << 31, "copy too large for one launch");
        switch (variant) {
        case 1: GAOT_KLAUNCH((k_stream_copy<1, 8>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 2: GAOT_KLAUNCH((k_stream_copy<2, 8>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 3: GAOT_KLAUNCH((k_stream_copy<3, 8>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 4: GAOT_KLAUNCH((k_stream_copy<1, 4>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 5: GAOT_KLAUNCH((k_stream_copy<2, 4>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 6: GAOT_KLAUNCH((k_stream_copy<3, 4>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 7: GAOT_KLAUNCH((k_stream_copy<1, 16>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        case 8: GAOT_KLAUNCH((k_stream_copy<1, 8>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        default: GAOT_KLAUNCH((k_stream_copy<1, 4>), dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, chunk4); break;
        }
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_stream_copy(const void* src, void* dst, int64_t bytes, gaot_stream_t stream) {
    return gaot_stream_copy_ex(src, dst, bytes, stream_copy_variant, stream);
}

extern "C" int gaot_patchify(const float* src, float* dst, int B, int Dd, int Hh, int Ww, int P, int C, int to_tokens,
                             gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(B > 0 && Dd > 0 && Hh > 0 && Ww > 0 && P > 0 && C > 0, "bad shape");
    GAOT_CHECK_ARG(Dd % P == 0 && Hh % P == 0 && Ww % P == 0, "Dimensions must be divisible by patch size");
    GAOT_CHECK_ARG(C % 4 == 0, "channels must be a multiple of 4");
    GAOT_CHECK_ARG(src && dst, "null pointer");
    const int64_t total = (int64_t)B * Dd * Hh * Ww * (C / 4);
    GAOT_KLAUNCH(k_patchify, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, B, Dd, Hh, Ww, P,
                       C, to_tokens);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" size_t gaot_mse_workspace_bytes(void) { return sizeof(double) * 1024 + 64; }

extern "C" int gaot_mse_fwd(const float* pred, const float* target, int64_t n, float* loss, void* workspace,
                            size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n > 0, "empty input");
    GAOT_CHECK_ARG(pred && target && loss && workspace && workspace_bytes >= gaot_mse_workspace_bytes(), "bad argument");
    const int nb = (int)std::min<int64_t>(1024, ceil_div(n, 256));
    double* part = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    GAOT_KLAUNCH(k_mse_part, dim3(nb), dim3(256), 0, st, pred, target, n, part);
    GAOT_KLAUNCH(k_mse_final, dim3(1), dim3(64), 0, st, part, nb, 1.0 / (double)n, loss);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_mse_bwd(const float* pred, const float* target, int64_t n, const float* grad_loss, float* dpred,
                            gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(n > 0, "empty input");
    GAOT_CHECK_ARG(pred && target && grad_loss && dpred, "null pointer");
    GAOT_KLAUNCH(k_mse_bwd, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n, grad_loss,
                       (float)(2.0 / (double)n), dpred);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}


extern "C" int gaot_scale_mix_fwd(const float* const* xs, int num_scales, const float* logits, float* out,
                                  float* weights, int64_t n, int channels, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_scales >= 1 && num_scales <= MAX_SCALES, "1..8 scales supported");
    if (channels != 32) {
        gaot_set_error("gaot_scale_mix_fwd: channels %d unsupported (only 32)", channels);
        return GAOT_ERR_UNSUPPORTED;
    }
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(xs && logits && out && weights, "null pointer");
    ScalePtrs p{};
    for (int s = 0; s < num_scales; ++s) p.x[s] = xs[s];
    GAOT_KLAUNCH(k_scale_mix_fwd, dim3(blocks_for(n * 32)), dim3(256), 0, (hipStream_t)stream, p, num_scales, logits,
                       out, weights, n);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_scale_mix_bwd(const float* const* xs, int num_scales, const float* weights, const float* dout,
                                  float* const* dxs, float* dlogits, int64_t n, int channels, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_scales >= 1 && num_scales <= MAX_SCALES, "1..8 scales supported");
    if (channels != 32) {
        gaot_set_error("gaot_scale_mix_bwd: channels %d unsupported (only 32)", channels);
        return GAOT_ERR_UNSUPPORTED;
    }
    if (n == 0) return GAOT_OK;
    GAOT_CHECK_ARG(xs && weights && dout && dxs && dlogits, "null pointer");
    ScalePtrs p{};
    for (int s = 0; s < num_scales; ++s) { p.x[s] = xs[s]; p.dx[s] = dxs[s]; }
    GAOT_KLAUNCH(k_scale_mix_bwd, dim3(blocks_for(n * 32)), dim3(256), 0, (hipStream_t)stream, p, num_scales,
                       weights, dout, dlogits, n);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
