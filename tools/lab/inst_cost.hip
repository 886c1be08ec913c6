// Instruction issue cost lab (gfx950): ONE wave per SIMD (256 threads per workgroup, one workgroup per CU), a loop of 64 copies of
// one instruction on rotating registers (no dependent chain shorter than 8 instructions); cycles per instruction from s_memtime.
// hipcc --offload-arch=gfx950 -O2 inst_cost.hip -o inst_cost && ./inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// registers: v[16+2i : 17+2i] destinations (i = 0..7), v[40:41], v[42:43] sources, s[20:21] a lane mask, a[0:127] accumulators
#define DEFKERNEL(NAME, LINE)                                                                                   \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned long long* out, int iters, float x) {               \
        unsigned long long t0, t1;                                                                              \
        asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"        \
                     "v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"        \
                     "s_mov_b64 s[20:21], 0x5555\n" ::"v"(x) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21"); \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));                                         \
        for (int it = 0; it < iters; ++it) {                                                                    \
            asm volatile(BODY64(LINE)::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", \
                         "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "vcc", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", \
                         "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
                         "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
                         "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
                         "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"); \
        }                                                                                                       \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                              \
    }

#define S_(x) #x
#define S(x) S_(x)
// destination pair index i -> v[16+2i]
#define L_MUL(i) "v_mul_f32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_mul, L_MUL)
#define L_FMA(i) "v_fma_f32 v[" S(16 + 2 * i) "], v40, v41, v42\n"
DEFKERNEL(k_fma, L_FMA)
#define L_PKMUL(i) "v_pk_mul_f32 v[" S(16 + 2 * i) ":" S(17 + 2 * i) "], v[40:41], v[42:43]\n"
DEFKERNEL(k_pkmul, L_PKMUL)
#define L_PKFMA(i) "v_pk_fma_f32 v[" S(16 + 2 * i) ":" S(17 + 2 * i) "], v[40:41], v[42:43], v[44:45]\n"
DEFKERNEL(k_pkfma, L_PKFMA)
#define L_PKADD(i) "v_pk_add_f32 v[" S(16 + 2 * i) ":" S(17 + 2 * i) "], v[40:41], v[42:43]\n"
DEFKERNEL(k_pkadd, L_PKADD)
#define L_EXP(i) "v_exp_f32 v[" S(16 + 2 * i) "], v40\n"
DEFKERNEL(k_exp, L_EXP)
#define L_EXP16(i) "v_exp_f16 v[" S(16 + 2 * i) "], v40\n"
DEFKERNEL(k_exp16, L_EXP16)
#define L_CND64(i) "v_cndmask_b32_e64 v[" S(16 + 2 * i) "], v40, v41, s[20:21]\n"
DEFKERNEL(k_cnd64, L_CND64)
#define L_CND32(i) "v_cndmask_b32_e32 v[" S(16 + 2 * i) "], v40, v41, vcc\n"
DEFKERNEL(k_cnd32, L_CND32)
#define L_CMPSDWA(i) "v_cmp_ge_u32_sdwa s[" S(22 + 2 * i) ":" S(23 + 2 * i) "], v40, v41 src0_sel:WORD_1 src1_sel:DWORD\n"
DEFKERNEL(k_cmpsdwa, L_CMPSDWA)
#define L_CMP64(i) "v_cmp_ge_u32_e64 s[" S(22 + 2 * i) ":" S(23 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_cmp64, L_CMP64)
#define L_CMP32(i) "v_cmp_ge_u32_e32 vcc, v40, v41\n"
DEFKERNEL(k_cmp32, L_CMP32)
#define L_CVTPK(i) "v_cvt_pk_bf16_f32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_cvtpk, L_CVTPK)
#define L_DOT2C(i) "v_dot2c_f32_bf16 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_dot2c, L_DOT2C)
#define L_BFI(i) "v_bfi_b32 v[" S(16 + 2 * i) "], v40, v41, v42\n"
DEFKERNEL(k_bfi, L_BFI)
#define L_BITOP3(i) "v_bitop3_b32 v[" S(16 + 2 * i) "], v40, v41, v42 bitop3:0x96\n"
DEFKERNEL(k_bitop3, L_BITOP3)
#define L_PKSUBI16(i) "v_pk_sub_i16 v[" S(16 + 2 * i) "], v40, v41 clamp\n"
DEFKERNEL(k_pksubi16, L_PKSUBI16)
#define L_PKASHR(i) "v_pk_ashrrev_i16 v[" S(16 + 2 * i) "], 15, v40\n"
DEFKERNEL(k_pkashr, L_PKASHR)
#define L_XOR(i) "v_xor_b32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_xor, L_XOR)
#define L_AND(i) "v_and_b32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_and, L_AND)
#define L_PERM(i) "v_perm_b32 v[" S(16 + 2 * i) "], v40, v41, v42\n"
DEFKERNEL(k_perm, L_PERM)
#define L_ADD(i) "v_add_f32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_add, L_ADD)
#define L_PKMULF16(i) "v_pk_mul_f16 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_pkmulf16, L_PKMULF16)
#define L_PKFMAF16(i) "v_pk_fma_f16 v[" S(16 + 2 * i) "], v40, v41, v42\n"
DEFKERNEL(k_pkfmaf16, L_PKFMAF16)
#define L_FMAMIX(i) "v_fma_mix_f32 v[" S(16 + 2 * i) "], v40, v41, v42 op_sel_hi:[1,0,0]\n"
DEFKERNEL(k_fmamix, L_FMAMIX)
#define L_MULSDWA(i) "v_mul_f32_sdwa v[" S(16 + 2 * i) "], v40, v41 dst_sel:DWORD src0_sel:DWORD src1_sel:DWORD\n"
DEFKERNEL(k_mulsdwa, L_MULSDWA)
#define L_MULLIT(i) "v_mul_f32 v[" S(16 + 2 * i) "], 0x3f9d70a4, v41\n"
DEFKERNEL(k_mullit, L_MULLIT)
#define L_MULS(i) "v_mul_f32 v[" S(16 + 2 * i) "], s20, v41\n"
DEFKERNEL(k_muls, L_MULS)
#define L_CVTF16(i) "v_cvt_pkrtz_f16_f32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_cvtf16, L_CVTF16)
#define L_LOG(i) "v_log_f32 v[" S(16 + 2 * i) "], v40\n"
DEFKERNEL(k_log, L_LOG)
#define L_RCP(i) "v_rcp_f32 v[" S(16 + 2 * i) "], v40\n"
DEFKERNEL(k_rcp, L_RCP)
#define L_MED3(i) "v_med3_f32 v[" S(16 + 2 * i) "], v40, v41, v42\n"
DEFKERNEL(k_med3, L_MED3)
#define L_LSHLADD(i) "v_lshl_add_u32 v[" S(16 + 2 * i) "], v40, 3, v42\n"
DEFKERNEL(k_lshladd, L_LSHLADD)
#define L_MULLO(i) "v_mul_lo_u32 v[" S(16 + 2 * i) "], v40, v41\n"
DEFKERNEL(k_mullo, L_MULLO)
// MFMA back to back on 8 independent accumulators
#define L_MFMA(i) "v_mfma_f32_32x32x16_bf16 a[" S(16 * i) ":" S(16 * i + 15) "], v[40:43], v[44:47], a[" S(16 * i) ":" S(16 * i + 15) "]\n"
DEFKERNEL(k_mfma, L_MFMA)
#define L_MFMA16(i) "v_mfma_f32_16x16x32_bf16 a[" S(16 * i) ":" S(16 * i + 3) "], v[40:43], v[44:47], a[" S(16 * i) ":" S(16 * i + 3) "]\n"
DEFKERNEL(k_mfma16, L_MFMA16)
// one MFMA followed by N independent VALU instructions: does the MFMA cost issue cycles in the stream?
#define L_MFMA_MUL7(i) L_MFMA(i) L_MUL(0) L_MUL(1) L_MUL(2) L_MUL(3) L_MUL(4) L_MUL(5) L_MUL(6)
DEFKERNEL(k_mfma_mul7, L_MFMA_MUL7)
#define L_MUL8(i) L_MUL(0) L_MUL(1) L_MUL(2) L_MUL(3) L_MUL(4) L_MUL(5) L_MUL(6) L_MUL(7)
#define L_MFMA_MUL8(i) L_MFMA(i) L_MUL8(i)
DEFKERNEL(k_mfma_mul8, L_MFMA_MUL8)
#define L_MFMA_MUL16(i) L_MFMA(i) L_MUL8(i) L_MUL8(i)
DEFKERNEL(k_mfma_mul16, L_MFMA_MUL16)
#define L_MFMA_MUL4(i) L_MFMA(i) L_MUL(0) L_MUL(1) L_MUL(2) L_MUL(3)
DEFKERNEL(k_mfma_mul4, L_MFMA_MUL4)
#define L_MFMA_EXP4(i) L_MFMA(i) L_EXP(0) L_EXP(1) L_EXP(2) L_EXP(3)
DEFKERNEL(k_mfma_exp4, L_MFMA_EXP4)
#define L_MFMA_PKMUL8(i) L_MFMA(i) L_PKMUL(0) L_PKMUL(1) L_PKMUL(2) L_PKMUL(3) L_PKMUL(4) L_PKMUL(5) L_PKMUL(6) L_PKMUL(7)
DEFKERNEL(k_mfma_pkmul8, L_MFMA_PKMUL8)
#define L_PKSUB8(i) L_PKSUBI16(0) L_PKSUBI16(1) L_PKSUBI16(2) L_PKSUBI16(3) L_PKSUBI16(4) L_PKSUBI16(5) L_PKSUBI16(6) L_PKSUBI16(7)
#define L_MFMA_PKSUB8(i) L_MFMA(i) L_PKSUB8(i)
DEFKERNEL(k_mfma_pksub8, L_MFMA_PKSUB8)
#define L_CVTPK8(i) L_CVTPK(0) L_CVTPK(1) L_CVTPK(2) L_CVTPK(3) L_CVTPK(4) L_CVTPK(5) L_CVTPK(6) L_CVTPK(7)
#define L_MFMA_CVTPK8(i) L_MFMA(i) L_CVTPK8(i)
DEFKERNEL(k_mfma_cvtpk8, L_MFMA_CVTPK8)
#define L_XOR8(i) L_XOR(0) L_XOR(1) L_XOR(2) L_XOR(3) L_XOR(4) L_XOR(5) L_XOR(6) L_XOR(7)
#define L_MFMA_XOR8(i) L_MFMA(i) L_XOR8(i)
DEFKERNEL(k_mfma_xor8, L_MFMA_XOR8)
#define L_CND8(i) L_CND64(0) L_CND64(1) L_CND64(2) L_CND64(3) L_CND64(4) L_CND64(5) L_CND64(6) L_CND64(7)
#define L_MFMA_CND8(i) L_MFMA(i) L_CND8(i)
DEFKERNEL(k_mfma_cnd8, L_MFMA_CND8)
#define L_DOT8(i) L_DOT2C(0) L_DOT2C(1) L_DOT2C(2) L_DOT2C(3) L_DOT2C(4) L_DOT2C(5) L_DOT2C(6) L_DOT2C(7)
#define L_MFMA_DOT8(i) L_MFMA(i) L_DOT8(i)
DEFKERNEL(k_mfma_dot8, L_MFMA_DOT8)
#define L_FMA8(i) L_FMA(0) L_FMA(1) L_FMA(2) L_FMA(3) L_FMA(4) L_FMA(5) L_FMA(6) L_FMA(7)
#define L_MFMA_FMA8(i) L_MFMA(i) L_FMA8(i)
DEFKERNEL(k_mfma_fma8, L_MFMA_FMA8)
#define L_PKFMA8(i) L_PKFMA(0) L_PKFMA(1) L_PKFMA(2) L_PKFMA(3) L_PKFMA(4) L_PKFMA(5) L_PKFMA(6) L_PKFMA(7)
#define L_MFMA_PKFMA8(i) L_MFMA(i) L_PKFMA8(i)
DEFKERNEL(k_mfma_pkfma8, L_MFMA_PKFMA8)
#define L_EXP8(i) L_EXP(0) L_EXP(1) L_EXP(2) L_EXP(3) L_EXP(4) L_EXP(5) L_EXP(6) L_EXP(7)
#define L_MFMA_EXP8(i) L_MFMA(i) L_EXP8(i)
DEFKERNEL(k_mfma_exp8, L_MFMA_EXP8)
// the MFMA with its B operand in AGPRs and a VGPR destination (the attention tile loops' S product)
#define L_MFMA_AB(i) "v_mfma_f32_32x32x16_bf16 v[16:31], v[40:43], a[" S(16 * i) ":" S(16 * i + 3) "], 0\n"
#define L_MFMAAB_MUL8(i) L_MFMA_AB(i) "v_mul_f32 v32, v40, v41\n v_mul_f32 v33, v40, v41\n v_mul_f32 v34, v40, v41\n v_mul_f32 v35, v40, v41\n v_mul_f32 v36, v40, v41\n v_mul_f32 v37, v40, v41\n v_mul_f32 v38, v40, v41\n v_mul_f32 v39, v40, v41\n"
DEFKERNEL(k_mfmaab_mul8, L_MFMAAB_MUL8)
// DEPENDENT MFMAs (every one accumulates into a[0:15]) with n independent v_mul_f32 between them: what a k-step chain costs
#define L_MFMAD "v_mfma_f32_32x32x16_bf16 a[0:15], v[40:43], v[44:47], a[0:15]\n"
#define L_MUL4 "v_mul_f32 v32, v40, v41\n v_mul_f32 v33, v40, v41\n v_mul_f32 v34, v40, v41\n v_mul_f32 v35, v40, v41\n"
#define L_MFMAD_0(i) L_MFMAD
DEFKERNEL(k_mfmad_0, L_MFMAD_0)
#define L_MFMAD_4(i) L_MFMAD L_MUL4
DEFKERNEL(k_mfmad_4, L_MFMAD_4)
#define L_MFMAD_8(i) L_MFMAD L_MUL4 L_MUL4
DEFKERNEL(k_mfmad_8, L_MFMAD_8)
#define L_MFMAD_12(i) L_MFMAD L_MUL4 L_MUL4 L_MUL4
DEFKERNEL(k_mfmad_12, L_MFMAD_12)
#define L_MFMAD_16(i) L_MFMAD L_MUL4 L_MUL4 L_MUL4 L_MUL4
DEFKERNEL(k_mfmad_16, L_MFMAD_16)
// two chains alternating (a[0:15], a[16:31]) with 4 v_mul_f32 after each: a dependent MFMA is then two issues behind its predecessor
#define L_MFMAD2 "v_mfma_f32_32x32x16_bf16 a[16:31], v[40:43], v[44:47], a[16:31]\n"
#define L_MFMAD_ALT4(i) L_MFMAD L_MUL4 L_MFMAD2 L_MUL4
DEFKERNEL(k_mfmad_alt4, L_MFMAD_ALT4)
#define L_DSREAD(i) "ds_read_b128 v[" S(16 + 2 * i) ":" S(19 + 2 * i) "], v48\n"
#define L_DSREADTR(i) "ds_read_b64_tr_b16 v[" S(16 + 2 * i) ":" S(17 + 2 * i) "], v48\n"

typedef void (*kern_t)(unsigned long long*, int, float);
struct Case { const char* name; kern_t k; int per_line; };

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8);
    const int iters = 20000;
    std::vector<Case> cases = {
        {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_fma_f32", k_fma, 1}, {"v_mul_f32 (literal)", k_mullit, 1}, {"v_mul_f32 (sgpr)", k_muls, 1},
        {"v_mul_f32_sdwa", k_mulsdwa, 1},
        {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_add_f32", k_pkadd, 1},
        {"v_pk_mul_f16", k_pkmulf16, 1}, {"v_pk_fma_f16", k_pkfmaf16, 1}, {"v_fma_mix_f32", k_fmamix, 1},
        {"v_exp_f32", k_exp, 1}, {"v_exp_f16", k_exp16, 1}, {"v_log_f32", k_log, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64, 1}, {"v_cndmask_b32_e32 (vcc)", k_cnd32, 1},
        {"v_cmp_ge_u32_sdwa -> sgpr", k_cmpsdwa, 1}, {"v_cmp_ge_u32_e64 -> sgpr", k_cmp64, 1}, {"v_cmp_ge_u32_e32 -> vcc", k_cmp32, 1},
        {"v_cvt_pk_bf16_f32", k_cvtpk, 1}, {"v_cvt_pkrtz_f16_f32", k_cvtf16, 1}, {"v_dot2c_f32_bf16", k_dot2c, 1},
        {"v_bfi_b32", k_bfi, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_pk_sub_i16 clamp", k_pksubi16, 1}, {"v_pk_ashrrev_i16", k_pkashr, 1},
        {"v_xor_b32", k_xor, 1}, {"v_and_b32", k_and, 1}, {"v_perm_b32", k_perm, 1}, {"v_med3_f32", k_med3, 1}, {"v_lshl_add_u32", k_lshladd, 1},
        {"v_mul_lo_u32", k_mullo, 1},
        {"v_mfma_f32_32x32x16_bf16 (8 chains)", k_mfma, 1}, {"v_mfma_f32_16x16x32_bf16 (8 chains)", k_mfma16, 1},
        {"group: 1 MFMA32 + 4 v_mul", k_mfma_mul4, 1}, {"group: 1 MFMA32 + 7 v_mul", k_mfma_mul7, 1}, {"group: 1 MFMA32 + 8 v_mul", k_mfma_mul8, 1},
        {"group: 1 MFMA32 + 16 v_mul", k_mfma_mul16, 1}, {"group: 1 MFMA32 + 4 v_exp", k_mfma_exp4, 1}, {"group: 1 MFMA32 + 8 v_pk_mul_f32", k_mfma_pkmul8, 1},
        {"group: 1 MFMA32 + 8 v_pk_fma_f32", k_mfma_pkfma8, 1}, {"group: 1 MFMA32 + 8 v_pk_sub_i16", k_mfma_pksub8, 1},
        {"group: 1 MFMA32 + 8 v_cvt_pk_bf16_f32", k_mfma_cvtpk8, 1}, {"group: 1 MFMA32 + 8 v_xor_b32", k_mfma_xor8, 1},
        {"group: 1 MFMA32 + 8 v_cndmask_b32_e64", k_mfma_cnd8, 1}, {"group: 1 MFMA32 + 8 v_dot2c_f32_bf16", k_mfma_dot8, 1},
        {"group: 1 MFMA32 + 8 v_fma_f32", k_mfma_fma8, 1}, {"group: 1 MFMA32 + 8 v_exp_f32", k_mfma_exp8, 1},
        {"group: 1 MFMA32 (B in AGPRs, D in VGPRs) + 8 v_mul_f32", k_mfmaab_mul8, 1},
        {"group: DEPENDENT MFMA32 alone", k_mfmad_0, 1}, {"group: DEPENDENT MFMA32 + 4 v_mul", k_mfmad_4, 1}, {"group: DEPENDENT MFMA32 + 8 v_mul", k_mfmad_8, 1},
        {"group: DEPENDENT MFMA32 + 12 v_mul", k_mfmad_12, 1}, {"group: DEPENDENT MFMA32 + 16 v_mul", k_mfmad_16, 1},
        {"group: two dependent chains alternating, 4 v_mul after each MFMA (2 MFMAs per line)", k_mfmad_alt4, 1},
    };
    for (auto& c : cases) {
        c.k<<<256, 256>>>(d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        c.k<<<256, 256>>>(d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx9: report both the counter and the wall time per line
        printf("%-42s %8.3f ns per line (wall)   memtime ticks %llu -> %.3f ticks/line\n", c.name, ms * 1e6 / ((double)iters * 64), t, (double)t / ((double)iters * 64));
    }
    return 0;
}
