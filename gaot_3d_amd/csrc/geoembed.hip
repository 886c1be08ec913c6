// Statistical geometric embedding features (reference GeometricEmbedding.
// _compute_statistical_features_pyg, src/model/layers/geoembed.py:99-182): per query node, from its
// incident edges:  [N_i, mean dist, var dist, centroid - query (3), eigenvalues of the centred
// covariance + 1e-6 I (3, descending)];  rows without neighbours are zeroed, then every column is
// z-scored over ALL query rows (unbiased std; std < 1e-6 -> 1).
//
// The reference spends 6 scatter passes + a batched LAPACK eigvalsh; here: one sweep over the
// row-sorted neighbour list (8 lanes per row, two passes per row: centroid, then covariance, all
// sums in fp64), a cyclic Jacobi 3x3 eigen-solve in registers, a two-stage column reduction and a
// normalise pass.  Inputs are coordinates only (no autograd), so the result is cacheable per sample.
#include "common.h"

namespace {

constexpr int NF = 9;
constexpr int G = 8;  // lanes per query row

__device__ __forceinline__ double grp_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq) {
    if (fabs(apq) < 1e-300) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = 0.0;
    const double rp = c * arp - s * arq, rq = s * arp + c * arq;
    arp = rp;
    arq = rq;
}

__global__ void k_geo_raw(const float* __restrict__ src_pos, const float* __restrict__ q_pos,
                          const int* __restrict__ rowptr, const int* __restrict__ src_sorted, int64_t Q,
                          float* __restrict__ feat) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int gl = threadIdx.x % G;
    if (row >= Q) return;  // whole groups exit together (G divides the block size)
    const int b = rowptr[row], e = rowptr[row + 1];
    const int n = e - b;
    float out[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) out[i] = 0.f;
    if (n > 0) {
        const double qx = q_pos[row * 3 + 0], qy = q_pos[row * 3 + 1], qz = q_pos[row * 3 + 2];
        double sd = 0, sd2 = 0, sx = 0, sy = 0, sz = 0;
        for (int i = b + gl; i < e; i += G) {
            const int s = src_sorted[i];
            const float fx = src_pos[(int64_t)s * 3 + 0], fy = src_pos[(int64_t)s * 3 + 1], fz = src_pos[(int64_t)s * 3 + 2];
            // distance as the reference computes it: fp32 difference, fp32 norm
            const float dx = fx - (float)qx, dy = fy - (float)qy, dz = fz - (float)qz;
            const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
            sd += dist;
            sd2 += (double)dist * (double)dist;
            sx += fx; sy += fy; sz += fz;
        }
        sd = grp_sum(sd); sd2 = grp_sum(sd2); sx = grp_sum(sx); sy = grp_sum(sy); sz = grp_sum(sz);
        const double inv = 1.0 / (double)n;
        const double davg = sd * inv;
        double dvar = sd2 * inv - davg * davg;
        if (dvar < 0.0) dvar = 0.0;
        const double cx = sx * inv, cy = sy * inv, cz = sz * inv;
        double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
        for (int i = b + gl; i < e; i += G) {
            const int s = src_sorted[i];
            const double ux = (double)src_pos[(int64_t)s * 3 + 0] - cx, uy = (double)src_pos[(int64_t)s * 3 + 1] - cy,
                         uz = (double)src_pos[(int64_t)s * 3 + 2] - cz;
            cxx += ux * ux; cxy += ux * uy; cxz += ux * uz; cyy += uy * uy; cyz += uy * uz; czz += uz * uz;
        }
        cxx = grp_sum(cxx); cxy = grp_sum(cxy); cxz = grp_sum(cxz); cyy = grp_sum(cyy); cyz = grp_sum(cyz); czz = grp_sum(czz);
        double a00 = cxx * inv + 1e-6, a11 = cyy * inv + 1e-6, a22 = czz * inv + 1e-6;
        double a01 = cxy * inv, a02 = cxz * inv, a12 = cyz * inv;
#pragma unroll 1
        for (int sweep = 0; sweep < 5; ++sweep) {   // cyclic Jacobi on 3x3 converges quadratically: 5 sweeps reach fp64 round-off
            jacobi_rot(a00, a11, a01, a02, a12);
            jacobi_rot(a00, a22, a02, a01, a12);
            jacobi_rot(a11, a22, a12, a01, a02);
        }
        double l0 = a00, l1 = a11, l2 = a22, t;
        if (l0 < l1) { t = l0; l0 = l1; l1 = t; }
        if (l0 < l2) { t = l0; l0 = l2; l2 = t; }
        if (l1 < l2) { t = l1; l1 = l2; l2 = t; }
        out[0] = (float)n; out[1] = (float)davg; out[2] = (float)dvar;
        out[3] = (float)(cx - qx); out[4] = (float)(cy - qy); out[5] = (float)(cz - qz);
        out[6] = (float)l0; out[7] = (float)l1; out[8] = (float)l2;
    }
    if (gl == 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) feat[row * NF + i] = out[i];
    }
}

// ---- point-sharded form: additive moments per query row, combined across ranks, then finished ---------------------
// mom[row][12] = {N, sum d, sum d^2, sum u (3), sum u u^T (xx, xy, xz, yy, yz, zz)} with u = x - q in fp64: every entry is
// a plain sum over the row's edges, so the rows of a sample whose edges are spread over several GPUs are obtained by a
// SUM all-reduce of the per-rank moments (gaot_3d_amd/sharding.py) -- no rank needs the full geometry.
constexpr int NM = 12;
__global__ void k_geo_moments(const float* __restrict__ src_pos, const float* __restrict__ q_pos,
                              const int* __restrict__ rowptr, const int* __restrict__ src_sorted, int64_t Q,
                              double* __restrict__ mom) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int gl = threadIdx.x % G;
    if (row >= Q) return;
    const int b = rowptr[row], e = rowptr[row + 1];
    const float qxf = q_pos[row * 3 + 0], qyf = q_pos[row * 3 + 1], qzf = q_pos[row * 3 + 2];
    double m[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) m[i] = 0.0;
    // four of the lane's edges per trip: their ids, then their 12 coordinate words, are requested TOGETHER (the one-edge loop was a
    // chain of two dependent round trips per edge -- 488 edges per token at 8 M points: 3.9 ms); the sums are formed in the same
    // order as before, edge by edge, so the moments are bit-identical
    constexpr int U = 4;
    for (int i0 = b + gl; i0 < e; i0 += U * G) {
        // (unconditional loads on clamped indices: a guarded load compiles to a branch with its own wait -- four dependent round trips)
        int sidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sidx[u] = src_sorted[min(i0 + u * G, e - 1)];
        float f[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t o = (int64_t)sidx[u] * 3;
            f[u][0] = src_pos[o + 0]; f[u][1] = src_pos[o + 1]; f[u][2] = src_pos[o + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(f[u][0]), "+v"(f[u][1]), "+v"(f[u][2]));   // all twelve words requested HERE, not inside the guards below
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * G >= e) break;
            const float fx = f[u][0], fy = f[u][1], fz = f[u][2];
            const float dx = fx - qxf, dy = fy - qyf, dz = fz - qzf;          // fp32 difference and norm, as the reference
            const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
            const double ux = (double)fx - (double)qxf, uy = (double)fy - (double)qyf, uz = (double)fz - (double)qzf;
            m[1] += dist; m[2] += (double)dist * (double)dist;
            m[3] += ux; m[4] += uy; m[5] += uz;
            m[6] += ux * ux; m[7] += ux * uy; m[8] += ux * uz; m[9] += uy * uy; m[10] += uy * uz; m[11] += uz * uz;
        }
    }
#pragma unroll
    for (int i = 1; i < NM; ++i) m[i] = grp_sum(m[i]);
    m[0] = (double)(e - b);
    if (gl == 0) {
#pragma unroll
        for (int i = 0; i < NM; ++i) mom[row * NM + i] = m[i];
    }
}

__global__ void k_geo_from_moments(const double* __restrict__ mom, int64_t Q, float* __restrict__ feat) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= Q) return;
    const double* m = mom + row * NM;
    float out[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) out[i] = 0.f;
    const double n = m[0];
    if (n > 0.5) {
        const double inv = 1.0 / n;
        const double davg = m[1] * inv;
        double dvar = m[2] * inv - davg * davg;
        if (dvar < 0.0) dvar = 0.0;
        const double ux = m[3] * inv, uy = m[4] * inv, uz = m[5] * inv;     // centroid - query
        double a00 = m[6] * inv - ux * ux + 1e-6, a11 = m[9] * inv - uy * uy + 1e-6, a22 = m[11] * inv - uz * uz + 1e-6;
        double a01 = m[7] * inv - ux * uy, a02 = m[8] * inv - ux * uz, a12 = m[10] * inv - uy * uz;
#pragma unroll 1
        for (int sweep = 0; sweep < 5; ++sweep) {   // cyclic Jacobi on 3x3 converges quadratically: 5 sweeps reach fp64 round-off
            jacobi_rot(a00, a11, a01, a02, a12);
            jacobi_rot(a00, a22, a02, a01, a12);
            jacobi_rot(a11, a22, a12, a01, a02);
        }
        double l0 = a00, l1 = a11, l2 = a22, t;
        if (l0 < l1) { t = l0; l0 = l1; l1 = t; }
        if (l0 < l2) { t = l0; l0 = l2; l2 = t; }
        if (l1 < l2) { t = l1; l1 = l2; l2 = t; }
        out[0] = (float)n; out[1] = (float)davg; out[2] = (float)dvar;
        out[3] = (float)ux; out[4] = (float)uy; out[5] = (float)uz;
        out[6] = (float)l0; out[7] = (float)l1; out[8] = (float)l2;
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) feat[row * NF + i] = out[i];
}

// per-block partial column sums (sum, sum of squares) in double
__global__ void k_geo_colpart(const float* __restrict__ feat, int64_t Q, double* __restrict__ part) {
    __shared__ double sm[4][2 * NF];
    double s[NF], s2[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) { s[i] = 0; s2[i] = 0; }
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < Q; r += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const double v = feat[r * NF + i];
            s[i] += v;
            s2[i] += v * v;
        }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        s[i] = wave_sum_d(s[i]);
        s2[i] = wave_sum_d(s2[i]);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) { sm[threadIdx.x >> 6][i] = s[i]; sm[threadIdx.x >> 6][NF + i] = s2[i]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * NF)
        part[(int64_t)blockIdx.x * 2 * NF + threadIdx.x] =
            sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// stats[0..8] = mean, stats[9..17] = std (1 if std < 1e-6).  One block: 32 column lanes (18 used) x 8 part lanes, the
// eight partial sums of a column combined in a fixed order (was one thread per column walking all parts: 63 us)
__global__ void k_geo_colfinal(const double* __restrict__ part, int nparts, int64_t Q, float* __restrict__ stats) {
    __shared__ double sm[8][2 * NF];
    const int c = threadIdx.x & 31, ry = threadIdx.x >> 5;
    if (c < 2 * NF) {
        double s = 0;
        for (int p = ry; p < nparts; p += 8) s += part[p * 2 * NF + c];
        sm[ry][c] = s;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i >= NF) return;
    double s = 0, s2 = 0;
    for (int j = 0; j < 8; ++j) { s += sm[j][i]; s2 += sm[j][NF + i]; }
    const double mean = s / (double)Q;
    double var = (Q > 1) ? (s2 - (double)Q * mean * mean) / (double)(Q - 1) : NAN;  // torch.std of one row = nan
    if (var < 0) var = 0;
    float sd = (float)sqrt(var);
    if (sd < 1e-6f) sd = 1.f;
    stats[i] = (float)mean;
    stats[NF + i] = sd;
}

// the same in two steps, for rows that are spread over several ranks: column sums of the local rows, then (after the
// caller's SUM all-reduce) mean / std over all Q_total rows
__global__ void k_geo_colsums(const double* __restrict__ part, int nparts, double* __restrict__ sums) {
    __shared__ double sm[8][2 * NF];
    const int c = threadIdx.x & 31, ry = threadIdx.x >> 5;
    if (c < 2 * NF) {
        double s = 0;
        for (int p = ry; p < nparts; p += 8) s += part[p * 2 * NF + c];
        sm[ry][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2 * NF) {
        double s = 0;
        for (int j = 0; j < 8; ++j) s += sm[j][threadIdx.x];
        sums[threadIdx.x] = s;
    }
}
__global__ void k_geo_stats_from_sums(const double* __restrict__ sums, int64_t Q, float* __restrict__ stats) {
    const int i = threadIdx.x;
    if (i >= NF) return;
    const double mean = sums[i] / (double)Q;
    double var = (Q > 1) ? (sums[NF + i] - (double)Q * mean * mean) / (double)(Q - 1) : NAN;
    if (var < 0) var = 0;
    float sd = (float)sqrt(var);
    if (sd < 1e-6f) sd = 1.f;
    stats[i] = (float)mean;
    stats[NF + i] = sd;
}

__global__ void k_geo_normalize(float* __restrict__ feat, int64_t Q, const float* __restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * NF) return;
    const int c = (int)(i % NF);
    feat[i] = (feat[i] - stats[c]) / stats[NF + c];
}

}  // namespace

extern "C" size_t gaot_geoembed_stats_workspace_bytes(void) { return sizeof(double) * 256 * 2 * NF + sizeof(float) * 2 * NF + 64; }

extern "C" int gaot_geoembed_moments(const float* source_pos, const float* query_pos, const int32_t* rowptr_dst,
                                     const int32_t* src_sorted, int64_t num_queries, double* moments, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_queries >= 0, "negative size");
    if (num_queries == 0) return GAOT_OK;
    GAOT_CHECK_ARG(source_pos && query_pos && rowptr_dst && moments, "null pointer");
    GAOT_KLAUNCH(k_geo_moments, dim3((unsigned)ceil_div(num_queries * G, 256)), dim3(256), 0, (hipStream_t)stream, source_pos,
                       query_pos, rowptr_dst, src_sorted, num_queries, moments);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_geoembed_from_moments(const double* moments, int64_t num_queries, float* features, void* workspace,
                                          size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_queries >= 0, "negative size");
    if (num_queries == 0) return GAOT_OK;
    GAOT_CHECK_ARG(moments && features && workspace, "null pointer");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_geoembed_stats_workspace_bytes(), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* stats = (float*)(part + 256 * 2 * NF);
    GAOT_KLAUNCH(k_geo_from_moments, dim3((unsigned)ceil_div(num_queries, 256)), dim3(256), 0, st, moments, num_queries, features);
    const int nb = (int)std::min<int64_t>(256, ceil_div(num_queries, 256));
    GAOT_KLAUNCH(k_geo_colpart, dim3(nb), dim3(256), 0, st, features, num_queries, part);
    GAOT_KLAUNCH(k_geo_colfinal, dim3(1), dim3(256), 0, st, part, nb, num_queries, stats);
    GAOT_KLAUNCH(k_geo_normalize, dim3((unsigned)ceil_div(num_queries * NF, 256)), dim3(256), 0, st, features,
                       num_queries, stats);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

// Two-sweep form (centroid first, then centred second moments) for queries that are spread over several ranks (decoder side of a point-sharded
// sample: every rank owns all edges of ITS queries, only the column z-score runs over all queries):
//   raw      : un-normalised features of the local queries + their column sums / sums of squares (18 doubles)
//   finalize : z-score with the (all-reduced) sums over num_queries_total rows
extern "C" int gaot_geoembed_raw(const float* source_pos, const float* query_pos, const int32_t* rowptr_dst,
                                 const int32_t* src_sorted, int64_t num_queries, float* features, double* colsums,
                                 void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_queries >= 0, "negative size");
    GAOT_CHECK_ARG(colsums && workspace, "null pointer");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_geoembed_stats_workspace_bytes(), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    if (num_queries == 0) {
        if (hipMemsetAsync(colsums, 0, sizeof(double) * 2 * NF, st) != hipSuccess) return GAOT_ERR_LAUNCH;
        return GAOT_OK;
    }
    GAOT_CHECK_ARG(source_pos && query_pos && rowptr_dst && features, "null pointer");
    GAOT_KLAUNCH(k_geo_raw, dim3((unsigned)ceil_div(num_queries * G, 256)), dim3(256), 0, st, source_pos, query_pos,
                       rowptr_dst, src_sorted, num_queries, features);
    const int nb = (int)std::min<int64_t>(256, ceil_div(num_queries, 256));
    GAOT_KLAUNCH(k_geo_colpart, dim3(nb), dim3(256), 0, st, features, num_queries, part);
    GAOT_KLAUNCH(k_geo_colsums, dim3(1), dim3(256), 0, st, part, nb, colsums);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}

extern "C" int gaot_geoembed_finalize(float* features, int64_t num_queries, const double* colsums, int64_t num_queries_total,
                                      void* workspace, size_t workspace_bytes, gaot_stream_t stream) {
    GAOT_ENTER();
    GAOT_CHECK_ARG(num_queries >= 0 && num_queries_total >= num_queries, "bad sizes");
    if (num_queries == 0) return GAOT_OK;
    GAOT_CHECK_ARG(features && colsums && workspace, "null pointer");
    GAOT_CHECK_ARG(workspace_bytes >= gaot_geoembed_stats_workspace_bytes(), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* stats = (float*)((double*)workspace + 256 * 2 * NF);
    GAOT_KLAUNCH(k_geo_stats_from_sums, dim3(1), dim3(64), 0, st, colsums, num_queries_total, stats);
    GAOT_KLAUNCH(k_geo_normalize, dim3((unsigned)ceil_div(num_queries * NF, 256)), dim3(256), 0, st, features,
                       num_queries, stats);
    GAOT_LAUNCH_CHECK();
    return GAOT_OK;
}
