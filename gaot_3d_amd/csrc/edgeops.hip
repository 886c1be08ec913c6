// General (unfused) per-edge operators for the variants of the GNO layers that the fused kernels (gno.hip,
// gno_bf16.hip, geoembed.hip) do not cover: IntegralTransform transform_type "nonlinear" / "nonlinear_kernelonly",
// segment-softmax attention weights (reference src/model/layers/integral_transform.py:68-78, 126-160), kernel MLPs
// of other widths, and the PointNet GeometricEmbedding (src/model/layers/geoembed.py:184-222).  The per-edge MLP
// itself runs through the GEMM kernels (gemm.hip / gemm_bf16.hip); what lives here is the gather / segment /
// element-wise glue and its autograd, all HBM-bound streaming kernels:
//   * every per-edge tensor is kept in the dst-sorted order of gaot_csr_build, so a segment is a contiguous row range;
//   * reductions run in a fixed order per row (no atomics) -> bit-reproducible, like the fused path;
//   * a reduction over the OTHER endpoint (gradient of a source-row gather) walks the src-sorted CSR through a
//     position map (src order -> dst order).
#include "common.h"

namespace {

constexpr int TPB = 256;

// out[e][c] = table[idx[e]][c]
__global__ void k_gather_rows(const float* __restrict__ table, int64_t ld, const int* __restrict__ idx, int64_t E, int C,
                              float* __restrict__ out, int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * C) return;
    const int64_t e = i / C;
    const int c = (int)(i - e * C);
    out[e * ldo + c] = table[(int64_t)idx[e] * ld + c];
}

// out[r][c] = reduce_{j in [rowptr[r], rowptr[r+1])} vals[map ? map[j] : j][c] ; mode 0 sum, 1 mean, 2 max (empty -> 0)
// one thread per (row, channel): the threads of a row read consecutive channels of the same edge (coalesced)
__global__ void k_segment_reduce(const float* __restrict__ vals, int64_t ldv, const int* __restrict__ rowptr,
                                 const int* __restrict__ map, int64_t R, int C, int mode, float* __restrict__ out,
                                 int* __restrict__ argmax) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const int lo = rowptr[r], hi = rowptr[r + 1];
    if (mode >= 2) {   // 2: max, 3: min (first extremum wins; empty row -> 0, scatter_native.py:32-50)
        float best = 0.f;
        int arg = -1;
        for (int j = lo; j < hi; ++j) {
            const int e = map ? map[j] : j;
            const float v = vals[(int64_t)e * ldv + c];
            if (arg < 0 || (mode == 2 ? v > best : v < best)) { best = v; arg = e; }
        }
        out[i] = best;
        if (argmax) argmax[i] = arg;
        return;
    }
    float s = 0.f;
    for (int j = lo; j < hi; ++j) {
        const int e = map ? map[j] : j;
        s += vals[(int64_t)e * ldv + c];
    }
    if (mode == 1) s /= (float)max(hi - lo, 1);
    out[i] = s;
}

// d_vals[argmax[r][c]][c] = d_out[r][c]  (d_vals pre-zeroed; targets are distinct: a row's edges belong to it alone)
__global__ void k_segment_max_bwd(const float* __restrict__ d_out, const int* __restrict__ argmax, int64_t R, int C,
                                  float* __restrict__ d_vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const int a = argmax[i];
    if (a >= 0) d_vals[(int64_t)a * C + (i % C)] = d_out[i];
}

// d_vals[e][c] = d_out[key[e]][c] * (mean ? 1/max(deg,1) : 1)   (backward of sum / mean over contiguous segments)
__global__ void k_segment_bcast(const float* __restrict__ d_out, const int* __restrict__ key, const int* __restrict__ rowptr,
                                int64_t E, int C, int mean, float* __restrict__ d_vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * C) return;
    const int64_t e = i / C;
    const int c = (int)(i - e * C);
    const int r = key[e];
    float v = d_out[(int64_t)r * C + c];
    if (mean) v /= (float)max(rowptr[r + 1] - rowptr[r], 1);
    d_vals[i] = v;
}

// segment softmax over contiguous segments: one wave per row
__global__ void k_segment_softmax_fwd(const float* __restrict__ s, const int* __restrict__ rowptr, int64_t R,
                                      float* __restrict__ w) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const int lo = rowptr[r], hi = rowptr[r + 1];
    float mx = -INFINITY;
    for (int j = lo + lane; j < hi; j += 64) mx = fmaxf(mx, s[j]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lo + lane; j < hi; j += 64) sum += __expf(s[j] - mx);
    // fixed-order combine of the 64 partial sums (butterfly: same order for every run)
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
    sum = fmaxf(sum, 1.17549435e-38f);   // clamp(min=tiny), integral_transform.py:76
    for (int j = lo + lane; j < hi; j += 64) w[j] = __expf(s[j] - mx) / sum;
}
// ds = w * (dw - sum_seg(w * dw))
__global__ void k_segment_softmax_bwd(const float* __restrict__ w, const float* __restrict__ dw,
                                      const int* __restrict__ rowptr, int64_t R, float* __restrict__ ds) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const int lo = rowptr[r], hi = rowptr[r + 1];
    float dot = 0.f;
    for (int j = lo + lane; j < hi; j += 64) dot += w[j] * dw[j];
    for (int o = 32; o; o >>= 1) dot += __shfl_xor(dot, o, 64);
    for (int j = lo + lane; j < hi; j += 64) ds[j] = w[j] * (dw[j] - dot);
}

// per-edge coordinate features; mode 0: out[e][0:6] = [y[src], x[dst]]; 1: out[e][0:3] = y[src] - x[dst];
// 2: out[e] = cos(x[dst], y[src]) with F.normalize semantics (v / max(|v|, 1e-12))
__global__ void k_edge_coords(const float* __restrict__ y, const float* __restrict__ x, const int* __restrict__ src,
                              const int* __restrict__ dst, int64_t E, int mode, float* __restrict__ out, int64_t ldo) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float* yp = y + 3 * (int64_t)src[e];
    const float* xp = x + 3 * (int64_t)dst[e];
    const float y0 = yp[0], y1 = yp[1], y2 = yp[2], x0 = xp[0], x1 = xp[1], x2 = xp[2];
    float* o = out + e * ldo;
    if (mode == 0) {
        o[0] = y0; o[1] = y1; o[2] = y2; o[3] = x0; o[4] = x1; o[5] = x2;
    } else if (mode == 1) {
        o[0] = y0 - x0; o[1] = y1 - x1; o[2] = y2 - x2;
    } else {
        const float ny = fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), 1e-12f);
        const float nx = fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), 1e-12f);
        o[0] = (x0 / nx) * (y0 / ny) + (x1 / nx) * (y1 / ny) + (x2 / nx) * (y2 / ny);
    }
}

// out = a .* b ; b is [rows][C] (b_row == 0) or one scalar per row (b_row == 1)
__global__ void k_mul(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int C, int b_row,
                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] * (b_row ? b[i / C] : b[i]);
}
// out[row] = sum_c a[row][c] * b[row][c]   (one wave per row, fixed-order butterfly)
__global__ void k_mul_rowsum(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int C,
                             float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += a[r * C + c] * b[r * C + c];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[r] = s;
}

// out[r][c] = x[r][c] * (1 + scale_m1[c]) + bias[c]   (time-conditioned norm, reference mlp.py:112-128; bias may be NULL)
__global__ void k_affine_cols(const float* __restrict__ x, const float* __restrict__ sm1, const float* __restrict__ bias,
                              int64_t n, int C, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    out[i] = fmaf(x[i], 1.0f + sm1[c], bias ? bias[c] : 0.f);
}

}  // namespace

extern "C" int gaot_affine_cols(const float* x, const float* scale_minus_one, const float* bias, int64_t rows, int C,
                                float* out, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && C > 0, "bad shape");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(x && scale_minus_one && out, "null pointer");
    GAOT_KLAUNCH(k_affine_cols, dim3((unsigned)ceil_div(rows * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, x,
                       scale_minus_one, bias, rows * C, C, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_gather_rows(const float* table, int64_t ld, const int* idx, int64_t E, int C, float* out, int64_t ldo,
                                gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(E >= 0 && C > 0 && ld >= C && ldo >= C, "bad shape");
    if (E == 0) return GAOT_OK;
    GAOT_CHECK_ARG(table && idx && out, "null pointer");
    GAOT_KLAUNCH(k_gather_rows, dim3((unsigned)ceil_div(E * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, table, ld, idx,
                       E, C, out, ldo);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_reduce(const float* vals, int64_t ldv, const int* rowptr, const int* map, int64_t R, int C,
                                   int mode, float* out, int* argmax, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(R >= 0 && C > 0 && ldv >= C && mode >= 0 && mode <= 3, "bad shape / mode");
    if (R == 0) return GAOT_OK;
    GAOT_CHECK_ARG(rowptr && out, "null pointer");
    GAOT_KLAUNCH(k_segment_reduce, dim3((unsigned)ceil_div(R * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, vals, ldv,
                       rowptr, map, R, C, mode, out, argmax);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_reduce_bwd(const float* d_out, const int* key, const int* rowptr, const int* argmax, int64_t R,
                                       int64_t E, int C, int mode, float* d_vals, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(R >= 0 && E >= 0 && C > 0 && mode >= 0 && mode <= 3, "bad shape / mode");
    if (E == 0) return GAOT_OK;
    GAOT_CHECK_ARG(d_out && d_vals && rowptr, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (mode >= 2) {
        GAOT_CHECK_ARG(argmax, "max / min need the arg-extremum of the forward");
        if (hipMemsetAsync(d_vals, 0, sizeof(float) * (size_t)E * C, st) != hipSuccess) {
            gaot_set_error("gaot_segment_reduce_bwd: memset failed");
            return GAOT_ERR_LAUNCH;
        }
        if (R > 0)
            GAOT_KLAUNCH(k_segment_max_bwd, dim3((unsigned)ceil_div(R * C, TPB)), dim3(TPB), 0, st, d_out, argmax, R, C, d_vals);
    } else {
        GAOT_CHECK_ARG(key, "sum / mean need the row of every edge");
        GAOT_KLAUNCH(k_segment_bcast, dim3((unsigned)ceil_div(E * C, TPB)), dim3(TPB), 0, st, d_out, key, rowptr, E, C,
                           mode == 1, d_vals);
    }
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_softmax_fwd(const float* scores, const int* rowptr, int64_t R, float* w, gaot_stream_t stream) {
    GAOT_ENTER();
    if (R <= 0) return GAOT_OK;
    GAOT_CHECK_ARG(rowptr && w, "null pointer");
    GAOT_KLAUNCH(k_segment_softmax_fwd, dim3((unsigned)ceil_div(R, TPB / 64)), dim3(TPB), 0, (hipStream_t)stream, scores,
                       rowptr, R, w);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_segment_softmax_bwd(const float* w, const float* dw, const int* rowptr, int64_t R, float* ds,
                                        gaot_stream_t stream) {
    GAOT_ENTER();
    if (R <= 0) return GAOT_OK;
    GAOT_CHECK_ARG(rowptr && ds, "null pointer");
    GAOT_KLAUNCH(k_segment_softmax_bwd, dim3((unsigned)ceil_div(R, TPB / 64)), dim3(TPB), 0, (hipStream_t)stream, w, dw,
                       rowptr, R, ds);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_edge_coords(const float* y_pos, const float* x_pos, const int* src, const int* dst, int64_t E, int mode,
                                float* out, int64_t ldo, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(E >= 0 && mode >= 0 && mode <= 2, "bad shape / mode");
    GAOT_CHECK_ARG(ldo >= (mode == 0 ? 6 : mode == 1 ? 3 : 1), "ldo too small for the mode");
    if (E == 0) return GAOT_OK;
    GAOT_CHECK_ARG(y_pos && x_pos && src && dst && out, "null pointer");
    GAOT_KLAUNCH(k_edge_coords, dim3((unsigned)ceil_div(E, TPB)), dim3(TPB), 0, (hipStream_t)stream, y_pos, x_pos, src,
                       dst, E, mode, out, ldo);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_mul(const float* a, const float* b, int64_t rows, int C, int b_is_row_scalar, float* out,
                        gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && C > 0, "bad shape");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(a && b && out, "null pointer");
    GAOT_KLAUNCH(k_mul, dim3((unsigned)ceil_div(rows * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, a, b, rows * C, C,
                       b_is_row_scalar, out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_mul_rowsum(const float* a, const float* b, int64_t rows, int C, float* out, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(rows >= 0 && C > 0, "bad shape");
    if (rows == 0) return GAOT_OK;
    GAOT_CHECK_ARG(a && b && out, "null pointer");
    GAOT_KLAUNCH(k_mul_rowsum, dim3((unsigned)ceil_div(rows, TPB / 64)), dim3(TPB), 0, (hipStream_t)stream, a, b, rows, C,
                       out);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
