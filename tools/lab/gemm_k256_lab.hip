// Lab: C[M,N] = A[M,256] . W[N,256]^T with the weight slice of a wave held in REGISTERS for the whole launch and the
// activation rows streamed through LDS by LDS-DMA (global_load_lds_dwordx4).  Standalone: hipcc --offload-arch=gfx950
// -O3 -o gemm_k256_lab gemm_k256_lab.hip -ldl ; ./gemm_k256_lab [path to libgaot3d_hip.so]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)__builtin_bit_cast(bf16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)b) << 16);
}

#ifndef ABL
#define ABL 0
#endif
constexpr int KK = 256, RB = 64, STAGE = RB * KK * 2;   // 32 KB per staged row block

template <bool C16, int OCC>
__global__ __launch_bounds__(256, OCC) void k_gemm_k256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, void* __restrict__ C,
                                                        int M, int N, int lda, int ldw, int ldc, int P, int subs) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, panel = j % P, sub = j / P;
    const int nblk = (M + RB - 1) / RB, c = xcd * subs + sub, nch = 8 * subs;
    const int t0 = (int)((int64_t)c * nblk / nch), t1 = (int)((int64_t)(c + 1) * nblk / nch);
    const int n0 = panel * 256 + wave * 64;

    bf16x8 bw[2][16];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        int n = n0 + 32 * jt + l31;
        n = n < N ? n : N - 1;
        const bf16_t* p = W + (int64_t)n * ldw + 8 * hf;
#pragma unroll
        for (int s = 0; s < 16; ++s) bw[jt][s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
    }

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((int64_t)M * lda * 2 > 0x7fffffff ? 0x7fffffff : (int64_t)M * lda * 2), 0x00020000);
    auto stage = [&](int t, int buf) {
        const int lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // LDS-DMA piece q = wave*8 + i covers tile rows 2q, 2q+1 (lane>>5 picks the row, lane&31 the 16-byte slot)
            const int x = 2 * i + lh;                                    // == row & 15
            const int voff = (wave * 16 + x) * lda * 2 + (((lane & 31) ^ x) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (wave * 8 + i) * 1024), 16, voff,
                                                 t * RB * lda * 2, 0, 0);
        }
    };

    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((int64_t)M * ldc * (C16 ? 2 : 4) > 0x7fffffff ? 0x7fffffff : (int64_t)M * ldc * (C16 ? 2 : 4)), 0x00020000);
    if (t0 < t1) stage(t0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        __builtin_amdgcn_s_barrier();
        if (t + 1 < t1) stage(t + 1, buf ^ 1);
        f32x16 acc[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
        const char* base = lds + buf * STAGE + l31 * 512;
        const int sw = l31 & 15;
        bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base + ((hf ^ sw) << 4));
        bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + 32 * 512 + ((hf ^ sw) << 4));
#pragma unroll
        for (int s = 0; s < ((ABL & 2) ? 1 : 16); ++s) {
            bf16x8 b0 = a0, b1 = a1;
            if (s + 1 < 16) {
                a0 = *reinterpret_cast<const bf16x8*>(base + (((2 * s + 2 + hf) ^ sw) << 4));
                a1 = *reinterpret_cast<const bf16x8*>(base + 32 * 512 + (((2 * s + 2 + hf) ^ sw) << 4));
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][s], b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[1][s], b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][s], b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[1][s], b1, acc[1][1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = t * RB + 32 * i + l31;
            const int esz = C16 ? 2 : 4;
            unsigned rowoff = m < M ? (unsigned)m * (unsigned)ldc * esz : 0x80000000u;
            if (ABL & 1) { if (acc[0][i][0] != 12345.678f) rowoff = 0x80000000u; }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                if constexpr (C16) {
                    unsigned pk[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pk[q][0] = pack2(acc[jt][i][4 * q], acc[jt][i][4 * q + 1]);
                        pk[q][1] = pack2(acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {   // pair (q, q+2): lower lanes end up with n = 8q..8q+7, upper lanes with n = 16+8q..
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 2][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 2][1], false, false);
                        u32x4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                        const int n = n0 + 32 * jt + 8 * q + 16 * hf;
                        __builtin_amdgcn_raw_buffer_store_b128(v, crs, rowoff + n * 2, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + 32 * jt + 8 * q + 4 * hf;
                        f32x4 v = {acc[jt][i][4 * q], acc[jt][i][4 * q + 1], acc[jt][i][4 * q + 2], acc[jt][i][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), crs, rowoff + n * 4, 0, 0);
                    }
                }
            }
        }
    }
}

__global__ void k_ref(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int lda, int ldw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < KK; ++k) {
        const float a = __uint_as_float((unsigned)A[(int64_t)m * lda + k] << 16), w = __uint_as_float((unsigned)W[(int64_t)n * ldw + k] << 16);
        s += a * w;
    }
    C[(int64_t)m * N + n] = s;
}

typedef int (*gemm_ex_t)(const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int, int,
                         const float*, int, const float*, int64_t, float*, int, void*, size_t, hipStream_t);

static bf16_t f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (bf16_t)(u >> 16);
}

template <bool C16, int OCC>
void run_new(const bf16_t* A, const bf16_t* W, void* C, int M, int N, int subs_div, hipStream_t st) {
    const int P = (N + 255) / 256;
    int subs = (64 / subs_div) / P;
    if (subs < 1) subs = 1;
    auto kern = k_gemm_k256<C16, OCC>;
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
        set = true;
    }
    hipLaunchKernelGGL(kern, dim3(8 * P * subs), dim3(256), 2 * STAGE, st, A, W, C, M, N, KK, KK, N, P, subs);
}

int main(int argc, char** argv) {
    const int M = 16384;
    gemm_ex_t gemm_ex = nullptr;
    if (argc > 1) {
        void* h = dlopen(argv[1], RTLD_NOW);
        if (!h) printf("dlopen failed: %s\n", dlerror());
        else gemm_ex = (gemm_ex_t)dlsym(h, "gaot_gemm_ex");
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int Ns[4] = {2048, 768, 256, 1024};
    const bool c16s[4] = {true, false, false, true};
    const int onlyN = argc > 2 ? atoi(argv[2]) : 0, onlyV = argc > 3 ? atoi(argv[3]) : -1;
    for (int cs = 0; cs < 4; ++cs) {
        const int N = Ns[cs];
        if (onlyN && N != onlyN) continue;
        const bool c16 = c16s[cs];
        std::vector<bf16_t> hA((size_t)M * KK), hW((size_t)N * KK);
        srand(1 + cs);
        for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : hW) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        const int NB = 3;
        bf16_t* dA[NB];
        void* dC[NB];
        bf16_t* dW;
        float* dRef;
        for (int i = 0; i < NB; ++i) {
            CK(hipMalloc(&dA[i], hA.size() * 2));
            CK(hipMemcpy(dA[i], hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
            CK(hipMalloc(&dC[i], (size_t)M * N * 4));
        }
        CK(hipMalloc(&dW, hW.size() * 2));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dRef, (size_t)M * N * 4));
        hipLaunchKernelGGL(k_ref, dim3((N + 255) / 256, M), dim3(256), 0, st, dA[0], dW, dRef, M, N, KK, KK);
        CK(hipStreamSynchronize(st));
        std::vector<float> ref((size_t)M * N);
        CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));

        for (int variant = 0; variant < 4; ++variant) {
            // 0: library kernel; 1: new, 2 WG/CU worth of workgroups (512); 2: new, 256 workgroups; 3: new 1024 workgroups
            if (variant == 0 && !gemm_ex) continue;
            if (onlyV >= 0 && variant != onlyV) continue;
            auto launch = [&](int i) {
                if (variant == 0) {
                    int rc = gemm_ex(dA[i % NB], dW, dC[i % NB], M, N, KK, KK, KK, N, 0, 1, 1, 1, c16 ? 1 : 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr, 0, st);
                    if (rc) { printf("gemm_ex rc %d\n", rc); exit(1); }
                } else {
                    const int div = variant == 1 ? 1 : (variant == 2 ? 2 : 1);
                    if (variant == 3) {
                        // 4 workgroups per CU's worth of ids (more, shorter workgroups)
                        const int P = (N + 255) / 256;
                        int subs = 128 / P; if (subs < 1) subs = 1;
                        if (c16) { auto k = k_gemm_k256<true, 2>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
                            hipLaunchKernelGGL(k, dim3(8 * P * subs), dim3(256), 2 * STAGE, st, dA[i % NB], dW, dC[i % NB], M, N, KK, KK, N, P, subs); }
                        else { auto k = k_gemm_k256<false, 2>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
                            hipLaunchKernelGGL(k, dim3(8 * P * subs), dim3(256), 2 * STAGE, st, dA[i % NB], dW, dC[i % NB], M, N, KK, KK, N, P, subs); }
                    } else if (c16) run_new<true, 2>(dA[i % NB], dW, dC[i % NB], M, N, div, st);
                    else run_new<false, 2>(dA[i % NB], dW, dC[i % NB], M, N, div, st);
                }
            };
            CK(hipMemsetAsync(dC[0], 0xff, (size_t)M * N * (c16 ? 2 : 4), st));
            launch(0);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            // check
            double maxerr = 0, maxref = 0;
            if (c16) {
                std::vector<bf16_t> out((size_t)M * N);
                CK(hipMemcpy(out.data(), dC[0], out.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < out.size(); ++i) {
                    unsigned u = (unsigned)out[i] << 16; float f; memcpy(&f, &u, 4);
                    double e = fabs((double)f - ref[i]); if (!(e <= maxerr)) maxerr = e;
                    if (fabs(ref[i]) > maxref) maxref = fabs(ref[i]);
                }
            } else {
                std::vector<float> out((size_t)M * N);
                CK(hipMemcpy(out.data(), dC[0], out.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < out.size(); ++i) {
                    double e = fabs((double)out[i] - ref[i]); if (!(e <= maxerr)) maxerr = e;
                    if (fabs(ref[i]) > maxref) maxref = fabs(ref[i]);
                }
            }
            for (int i = 0; i < 5; ++i) launch(i);
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 60;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, tf = 2.0 * M * N * KK / (us * 1e-6) / 1e12;
            printf("N=%4d %s variant %d: %7.2f us  %6.1f TF/s   max|err| %.3e (peak %.2f)\n", N, c16 ? "bf16-out" : "fp32-out", variant, us, tf, maxerr, maxref);
        }
        for (int i = 0; i < NB; ++i) { CK(hipFree(dA[i])); CK(hipFree(dC[i])); }
        CK(hipFree(dW)); CK(hipFree(dRef));
    }
    return 0;
}
