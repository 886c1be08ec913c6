// Lab: is the scalar offset (soffset) of a raw buffer access part of the hardware range check on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* data, float* out, float* scratch) {
    // resource over the FIRST 16 floats (64 bytes) of data
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)data, 0, 64, 0x00020000);
    const int lane = threadIdx.x;
    // (a) voffset in range, soffset pushes the address past num_records
    out[0 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4 % 64, 128, 0));
    // (b) the same address through voffset alone
    out[1 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 128 + lane * 4 % 64, 0, 0));
    // (c) in range
    out[2 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4 % 64, 0, 0));
    // stores: (d) soffset past the end, (e) voffset past the end
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc((void*)scratch, 0, 64, 0x00020000);
    if (lane < 16) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 7.0f), ss, lane * 4, 128, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 9.0f), ss, 256 + lane * 4, 0, 0);
    }
}
int main() {
    float *d, *o, *s, h[256], ho[192], hs[256];
    for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    hipMalloc(&d, 1024); hipMalloc(&o, 768); hipMalloc(&s, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice); hipMemset(s, 0, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, s);
    hipMemcpy(ho, o, 768, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 1024, hipMemcpyDeviceToHost);
    printf("load  soffset past end : %g %g  (0 = range-checked, 132.. = NOT checked)\n", ho[0], ho[1]);
    printf("load  voffset past end : %g %g\n", ho[64], ho[65]);
    printf("load  in range         : %g %g\n", ho[128], ho[129]);
    printf("store soffset past end : scratch[32] = %g (7 = written: NOT checked)\n", hs[32]);
    printf("store voffset past end : scratch[64] = %g (9 = written)\n", hs[64]);
    return 0;
}
