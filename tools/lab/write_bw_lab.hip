// Lab (round 6): pure-store bandwidth of the chip: N workgroups each writing contiguous 16-KiB chunks (16 B per lane), over a buffer far
// larger than the 256 MiB Infinity Cache, plain and non-temporal; and the same bytes as a copy (read + write) for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void k_fill(f32x4* __restrict__ dst, long n16) {
    const long chunk = 1024;   // 16-byte elements per workgroup chunk = 16 KiB
    for (long c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        f32x4* p = dst + c * chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = {1.f, 2.f, 3.f, (float)i};
            if (NT) __builtin_nontemporal_store(v, p + i * 256 + threadIdx.x);
            else p[i * 256 + threadIdx.x] = v;
        }
    }
}
__global__ void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n16) {
    const long chunk = 1024;
    for (long c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_nontemporal_load(src + c * chunk + i * 256 + threadIdx.x);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(v[i], dst + c * chunk + i * 256 + threadIdx.x);
    }
}
int main() {
    const long bytes = 2l << 30, n16 = bytes / 16;
    f32x4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 2; ++w) { if (mode == 0) k_fill<0><<<grid, 256>>>(a, n16); else if (mode == 1) k_fill<1><<<grid, 256>>>(a, n16); else k_copy<<<grid, 256>>>(b, a, n16); }
            CK(hipEventRecord(e0));
            const int reps = 5;
            for (int r = 0; r < reps; ++r) { if (mode == 0) k_fill<0><<<grid, 256>>>(a, n16); else if (mode == 1) k_fill<1><<<grid, 256>>>(a, n16); else k_copy<<<grid, 256>>>(b, a, n16); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("grid %d %s: %.2f TB/s of %s\n", grid, mode == 0 ? "fill" : mode == 1 ? "fill nt" : "copy nt", (double)bytes * reps * (mode == 2 ? 2 : 1) / (ms * 1e-3) / 1e12, mode == 2 ? "read + written bytes" : "written bytes");
        }
    }
    return 0;
}
