// Lab (round 6): the fused FFN forward kernel of csrc/ffn_fused.hip timed alone, with ablations chosen at compile time
// (-DGAOT_FFN_ABL=<bits>, -DGAOT_FFN_FWD_RING=<3|6>).  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I gaot_3d_amd/csrc -o ffn_fwd_lab tools/lab/ffn_fwd_lab.hip
#include "../../gaot_3d_amd/csrc/abi.hip"
#include "../../gaot_3d_amd/csrc/ffn_fused.hip"

#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, F = 1024;
    std::vector<float> w13((size_t)2 * F * 256), w2((size_t)256 * F);
    for (size_t i = 0; i < w13.size(); ++i) w13[i] = 0.06f * (float)((int)(i * 2654435761u % 2001) - 1000) / 1000.f;
    for (size_t i = 0; i < w2.size(); ++i) w2[i] = 0.03f * (float)((int)(i * 40503u % 2001) - 1000) / 1000.f;
    std::vector<unsigned short> x((size_t)M * 256);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (unsigned short)(0x3c00 + (i * 7919u % 512));   // bf16 values near 0.0078 .. 0.03
    float *dw13, *dw2, *dy, *dr;
    void *dx, *dp, *dag, *du;
    CK(hipMalloc(&dw13, w13.size() * 4)); CK(hipMalloc(&dw2, w2.size() * 4)); CK(hipMalloc(&dx, x.size() * 2));
    CK(hipMalloc(&dp, gaot_ffn_packed_bytes(F, 1))); CK(hipMalloc(&dy, (size_t)M * 256 * 4)); CK(hipMalloc(&dr, (size_t)M * 256 * 4));
    CK(hipMalloc(&dag, (size_t)M * 2 * F * 2)); CK(hipMalloc(&du, (size_t)M * F * 2));
    CK(hipMemcpy(dw13, w13.data(), w13.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(dr, 0, (size_t)M * 256 * 4));
    if (gaot_ffn_pack(dw13, dw2, F, dp, 1, nullptr) != 0) { printf("pack failed: %s\n", gaot_last_error()); return 1; }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int save = 0; save < 2; ++save) {
        for (int i = 0; i < 5; ++i)
            if (gaot_ffn_fwd(dx, dp, dr, 256, dy, save ? dag : nullptr, save ? du : nullptr, M, F, nullptr) != 0) { printf("fwd failed: %s\n", gaot_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        const int reps = 200;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) gaot_ffn_fwd(dx, dp, dr, 256, dy, save ? dag : nullptr, save ? du : nullptr, M, F, nullptr);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("ABL=%d RING=%d M=%d save=%d: %.1f us\n", GAOT_FFN_ABL, GAOT_FFN_FWD_RING, M, save, ms / reps * 1e3);
#ifdef GAOT_FFN_TIMING
        unsigned long long ts[64];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ffn_t), sizeof(ts)));
        printf("  cycles (s_memtime, 100 MHz ticks?) from start: tile+barrier %llu, prologue %llu;", ts[1] - ts[0], ts[2] - ts[1]);
        for (int c = 0; c < 8; ++c) printf(" c%d: merged %llu bar %llu y %llu;", c, ts[3 + 3 * c] - (c ? ts[2 + 3 * c] : ts[2]), ts[4 + 3 * c] - ts[3 + 3 * c], ts[5 + 3 * c] - ts[4 + 3 * c]);
        printf(" store %llu; total %llu\n", ts[40] - ts[26], ts[40] - ts[0]);
#endif
    }
    return 0;
}
