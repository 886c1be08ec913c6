// Backward of the fused GNO integral transform on the bf16 matrix cores, second design (precision 1 of gaot_gno_bwd;
// reference semantics: integral_transform.py:146-171 + LinearChannelMLP mlp.py:327-335, see gno.hip).
//
// What the first design (gno_bf16.hip: k_gno_bwd_bf16) spent its time on was waiting: 51 % of the wave time in s_waitcnt
// at one wave per SIMD -- every MFMA operand of the MLP was a 16-byte buffer_load from L2, the f[src] / g[dst] row gathers
// were issued right before their use, h / dz tiles were transposed into LDS with 2-byte stores, and the weight-gradient
// products of each layer sat between two workgroup barriers (8 barriers per 128 edges).  Here:
//   * the MLP's bf16 operand fragments (recompute AND transposed data-gradient forms, 40 KB for 3 hidden layers) live in
//     LDS for the whole launch: one conflict-free ds_read_b128 per MFMA operand;
//   * the gathered rows are requested at the top of the tile and consumed after the MLP recompute;
//   * activations h_l and pre-activation gradients dz_l of a tile are stored in their accumulator layout as
//     [edge][feature] 32x32 bf16 tiles (four 8-byte stores per block) and every weight-gradient operand is a hardware-
//     transposed read (ds_read_b64_tr_b16) of those tiles;
//   * ALL weight-gradient products of the four tiles run in ONE phase at the end of the iteration: two workgroup barriers
//     per 128 edges instead of eight; each wave owns one 32x32 output tile per hidden layer plus one of dW_L / dW_0
//     (48 accumulator registers), bias gradients ride on the same A fragments through a one-hot selector operand.
// Determinism as before: segmented sums in tile order, per-workgroup weight partials reduced in workgroup order.
#include <stdlib.h>

#include "gno_common.h"
#include "tile32.h"

namespace {

using namespace gno;

typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ constexpr int kmap(int s, int j, int hf) { return 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3); }

// like frag_cols (contract over tile ROWS, lane = column), but element j <-> row 16s + 8hf + j: the NATURAL k order of a
// row read (frag_rows), so that one operand may come from a row read and the other from a transposed read
__device__ __forceinline__ bf16x8 frag_cols_nat(const char* tile, int lane, int s) {
    const int i = lane & 15, grp = (lane >> 4) & 1, hf = lane >> 5;
    const int col = 16 * grp + 4 * (i & 3);
    const int r0 = 16 * s + 8 * hf + (i >> 2), r1 = r0 + 4;
    const char* p0 = tile + tile_off(r0, col >> 3) + ((col & 7) << 1);
    const char* p1 = tile + tile_off(r1, col >> 3) + ((col & 7) << 1);
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}

// the 16 fp32 values of an accumulator tile rounded to bf16 ONCE, as 8 packed pairs (v_cvt_pk_bf16_f32): the same words
// are the two operand fragments of the next product (acc_to_frags order) and the 8-byte pieces of the LDS tile
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
struct Packed16 { unsigned w[8]; };
__device__ __forceinline__ Packed16 pack16(const f32x16& v) {
    Packed16 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32v2 t = {v[2 * j], v[2 * j + 1]};
        p.w[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16v2));
    }
    return p;
}
__device__ __forceinline__ void frags_of(const Packed16& p, bf16x8& f0, bf16x8& f1) {
    f0 = __builtin_bit_cast(bf16x8, make_uint4(p.w[0], p.w[1], p.w[2], p.w[3]));
    f1 = __builtin_bit_cast(bf16x8, make_uint4(p.w[4], p.w[5], p.w[6], p.w[7]));
}
// accumulator tile (lane l31 = tile row, register r <-> tile column (r&3) + 8(r>>2) + 4hf) -> bf16 tile in LDS
__device__ __forceinline__ void store_packed_rows(char* blk, const Packed16& p, int l31, int hf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(blk + tile_off(l31, g) + 8 * hf) = make_uint2(p.w[2 * g], p.w[2 * g + 1]);
}
__device__ __forceinline__ void store_acc_rows(char* blk, const f32x16& v, int l31, int hf) { store_packed_rows(blk, pack16(v), l31, hf); }

// a * (f16 half of g).  Written as an fma with a +0 addend: the compiler cannot fold it to a multiply (signed zeros), so it
// selects v_fma_mix_f32, which reads the f16 half directly -- one instruction instead of a conversion and a multiply, and
// no inline asm (an asm statement that reads an MFMA result gets no hazard padding from the compiler)
__device__ __forceinline__ float mul_f16lo(float a, unsigned g) { return __builtin_fmaf(a, (float)__builtin_bit_cast(h16x2, g)[0], 0.0f); }
__device__ __forceinline__ float mul_f16hi(float a, unsigned g) { return __builtin_fmaf(a, (float)__builtin_bit_cast(h16x2, g)[1], 0.0f); }

template <int NH>
struct Lds2 {
    static constexpr int H = 64, C = 32, KB = 2;
    static constexpr int per_hidden = KB * KB * 2 * 64 * 16;   // bytes of one hidden layer's fragment image
    static constexpr int per_last = KB * 2 * 64 * 16;
    static constexpr int img_bytes = 2 * ((NH - 1) * per_hidden + per_last);   // fw[1..NH-1], fw[NH], bw[1..NH-1], bw[NH]
    static constexpr int fw(int l) { return (l - 1) * per_hidden; }            // l = 1..NH (NH = last)
    static constexpr int bw(int l) { return (NH - 1) * per_hidden + per_last + (l - 1) * per_hidden; }
    static constexpr int w0t = img_bytes;                       // float [8][64] (rows 6, 7 zero)
    static constexpr int bias = w0t + 8 * H * 4;                // float NH*64 + 32
    static constexpr int tiles = (bias + (NH * H + C) * 4 + 15) & ~15;
    static constexpr int h(int l) { return l * 4096; }          // per-wave offsets
    static constexpr int dz(int l) { return NH * 4096 + l * 4096; }
    static constexpr int dk = 2 * NH * 4096;
    static constexpr int in = dk + 2048;
    static constexpr int ids = in + 512;
    static constexpr int per_wave = ids + 256;
    static constexpr int total = tiles + 4 * per_wave;
};

template <int NH>
struct ParamLayout2 {  // flat per-workgroup partial layout, state_dict order (same as gno.hip / gno_bf16.hip)
    static constexpr int H = 64, C = 32;
    static constexpr int w_off(int l) { return l == 0 ? 0 : (H * IN0 + H) + (l - 1) * (H * H + H); }
    static constexpr int b_off(int l) { return w_off(l) + (l == 0 ? H * IN0 : (l == NH ? C * H : H * H)); }
    static constexpr int total = (H * IN0 + H) + (NH - 1) * (H * H + H) + (C * H + C);
};

// STAMP: diagnostic build (GAOT_GNO_STAMPS=1) that sums s_memtime differences per phase into a debug buffer; its fences
// forbid overlaps the real kernel has, so read the SHARES it prints, not its run time
#define GNO_STAMP(i)                                                                              \
    if constexpr (STAMP) {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        tacc[i] += t_ - tlast;                                                                    \
        tlast = t_;                                                                               \
    }

template <int NH, bool STAMP>
__global__ __launch_bounds__(256, 1) void k_gno_bwd2_bf16(
    const uint4* __restrict__ images, const float* __restrict__ w0t_g, MlpPtrs mlp, const float* __restrict__ y_pos,
    const float* __restrict__ x_pos, const float* __restrict__ f_y, const float* __restrict__ gs,
    const int* __restrict__ src_s, const int* __restrict__ dst_s, const int* __restrict__ rowptr_src, int64_t E,
    float* __restrict__ grad_f, float* __restrict__ part, float* __restrict__ wpart, unsigned long long* __restrict__ stamps) {
    constexpr int C = 32, H = 64, KB = 2;
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    using L = Lds2<NH>;
    using PL = ParamLayout2<NH>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;

    // ---- resident operands: fragment images (prepared by k_prep_bwd_images), layer-0 weight, biases -------------------
    for (int i = threadIdx.x; i < L::img_bytes / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = images[i];
    {
        float* w0 = reinterpret_cast<float*>(lds + L::w0t);
        for (int i = threadIdx.x; i < 8 * H; i += 256) w0[i] = (i < IN0 * H) ? w0t_g[i] : 0.f;
        float* bl = reinterpret_cast<float*>(lds + L::bias);
#pragma unroll
        for (int l = 0; l < NH; ++l)
            for (int i = threadIdx.x; i < H; i += 256) bl[l * H + i] = mlp.b[l][i];
        for (int i = threadIdx.x; i < C; i += 256) bl[NH * H + i] = mlp.b[NH][i];
    }
    __syncthreads();
    const float* w0 = reinterpret_cast<const float*>(lds + L::w0t);
    const float* bias_l = reinterpret_cast<const float*>(lds + L::bias);
    auto img = [&](int off_bytes, int frag) { return reinterpret_cast<const bf16x8*>(lds + off_bytes)[frag * 64 + lane]; };
    auto wave_base = [&](int w) { return lds + L::tiles + w * L::per_wave; };
    char* mine = wave_base(wave);
    char* dkT = mine + L::dk;
    char* inT = mine + L::in;

    const int wjb = wave >> 1, wkb = wave & 1;
    f32x16 dWh[NH > 1 ? NH - 1 : 1], dWx, bacc;   // dWx: dW_L tile (waves 0,1: k-block wave) / dW_0 tile (waves 2,3: j-block wave-2)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dWx[r] = 0.f; bacc[r] = 0.f; }
#pragma unroll
    for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWh[l][r] = 0.f;
    auto sel = [&](int col) {   // one-hot selector fragment: B[e][n] = (n == col)
        bf16x8 f;
        const short one = (l31 == col) ? (short)0x3F80 : (short)0;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = one;
        return f;
    };

    // the gathered tables as buffer resources (rows are 128 B, 2 GB per resource: tables of up to 2^24 rows; gaot_gno_bwd checks)
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gs, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)f_y, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rgf = __builtin_amdgcn_make_buffer_rsrc((void*)grad_f, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpart = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, 0x7fffffff, 0x00020000);
    const int64_t n_tiles = (E + 31) / 32;
    // ids and endpoint coordinates of a tile are two DEPENDENT global round trips: fetched one iteration ahead
    int s_nx = 0, q_nx = 0;
    bool v_nx = false;
    float bin_nx[3] = {0.f, 0.f, 0.f};
    auto fetch_ids = [&](int64_t tb_) {
        const int64_t e = (tb_ + wave) * 32 + l31;
        v_nx = tb_ < n_tiles && e < E;
        s_nx = v_nx ? src_s[e] : 0;
        q_nx = v_nx ? dst_s[e] : 0;
    };
    auto fetch_pos = [&]() {
        const float* ys = y_pos + (int64_t)s_nx * 3;
        const float* xq = x_pos + (int64_t)q_nx * 3;
        bin_nx[0] = ys[hf];
        bin_nx[1] = hf ? xq[0] : ys[2];
        bin_nx[2] = xq[1 + hf];
    };
    fetch_ids((int64_t)blockIdx.x * 4);
    fetch_pos();
    if constexpr (STAMP) tlast = __builtin_amdgcn_s_memtime();
    for (int64_t tb = (int64_t)blockIdx.x * 4; tb < n_tiles; tb += (int64_t)gridDim.x * 4) {
        const int64_t base = (tb + wave) * 32;
        // compiler-only memory barrier: without it the loop-invariant LDS reads of the operand images are hoisted out of the
        // tile loop into (and beyond) the whole register file -- 40 spilled registers reloaded from scratch per tile
        asm volatile("" ::: "memory");
        float bin[3];
        int idv, rbv = 0, rev = 0;   // per edge (lane l31, both halves alike): source row or -1, and its [rb, re) edge range
        int s_raw, q_raw;            // endpoints as fetched (0 for edges past E)
        {
            const bool valid = v_nx;
            const int s = s_nx, q = q_nx;
            bin[0] = bin_nx[0]; bin[1] = bin_nx[1]; bin[2] = bin_nx[2];
            fetch_ids(tb + (int64_t)gridDim.x * 4);
            idv = valid ? s : -1;
            s_raw = s;
            q_raw = q;
            if (valid) {
                rbv = rowptr_src[s];
                rev = rowptr_src[s + 1];
            }
            // input tile [k][e] bf16 (rows 0..5 = coordinates, row 6 = ones -> db_0, row 7 = 0)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                *reinterpret_cast<bf16_t*>(inT + tile_off(2 * i + hf, l31 >> 3) + ((l31 & 7) << 1)) = f2bf(bin[i]);
            *reinterpret_cast<bf16_t*>(inT + tile_off(6 + hf, l31 >> 3) + ((l31 & 7) << 1)) = hf ? (bf16_t)0 : (bf16_t)0x3F80;
        }
        GNO_STAMP(8)   // (diagnostic split of phase 0) prefetched ids / coordinates consumed, next ids + rowptr requested, input tile
        // ---- gather f[src] / g[dst] rows now: they are needed after the MLP recompute (lane = channel, reg = edge); the
        //      edge's endpoints come from the lane that holds the edge (ds_bpermute: no LDS round trip, no fence) ---------
        // Unconditional loads (edges past E carry endpoint 0, a valid row): a select on the loaded value would make the
        // compiler wait for every group of loads before issuing the next; the rows of such edges are zeroed at their use.
        float fv[16], gv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int el = mfma32_row(r, hf);
            const int s_ = __builtin_amdgcn_ds_bpermute(4 * el, s_raw), q_ = __builtin_amdgcn_ds_bpermute(4 * el, q_raw);
            // buffer loads: one 32-bit offset register per row instead of a 64-bit address pair
            gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, q_ * (C * 4) + l31 * 4, 0, 0));
            fv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, s_ * (C * 4) + l31 * 4, 0, 0));
        }
        GNO_STAMP(0)   // ids, input tile, gather issue
        // ---- recompute the MLP: gelu'(z) stays in registers (fp32), h_l goes to LDS as bf16 [edge][feature] ------------
        // gelu'(z) is kept as packed f16 pairs (in [-0.13, 1.13]: 11 significant bits against the 8 of the bf16 operands it
        // multiplies into): half the registers of the fp32 form, and the multiply reads the halves directly
        // ... and, for the layers above the first, parked in LDS: gelu'(z_l) of this wave's tile waits in the slot its own
        // dz_l tile will overwrite right after consuming it (same 4 KB, lane-linear 16-byte pieces).  Keeping 16 NH registers
        // live across the whole MLP pushed every MFMA result into AGPRs: ~380 v_accvgpr_read per tile, a quarter of the
        // VALU issue, plus 252 B of scratch for three hidden layers.  Layer 0 stays in registers: its slot dz(0) is the
        // scratch of the segmented sums in between.
        unsigned gp0[KB][8];
        auto gp_store = [&](int l, int ob, const unsigned (&w)[8]) {
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gp0[ob][j] = w[j];
                return;
            }
            char* p = mine + L::dz(l) + ob * 2048 + lane * 16;
            *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(p + 1024) = make_uint4(w[4], w[5], w[6], w[7]);
        };
        auto gp_load = [&](int l, int kb, unsigned (&w)[8]) {
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = gp0[kb][j];
                return;
            }
            const char* p = mine + L::dz(l) + kb * 2048 + lane * 16;
            const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 1024);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        bf16x8 hb[KB][2];
#pragma unroll
        for (int ob = 0; ob < KB; ++ob) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = bias_l[32 * ob + mfma32_row(r, hf)];
#pragma unroll
            for (int i = 0; i < IN0 / 2; ++i)
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(2 * i + hf) * H + 32 * ob + l31], bin[i], z, 0, 0, 0);
            unsigned gw[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32v2 g, d;
                gelu_fast_pair2(f32v2{z[r], z[r + 1]}, g, d);
                z[r] = g[0];
                z[r + 1] = g[1];
                gw[r >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]));
            }
            gp_store(0, ob, gw);
            const Packed16 pz = pack16(z);
            store_packed_rows(mine + L::h(0) + ob * TILE_BYTES, pz, l31, hf);
            frags_of(pz, hb[ob][0], hb[ob][1]);
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            // keep the scheduler from pulling the next layers' operand reads up here: with one wave per SIMD it would
            // trade registers for latency it does not need to hide and run out of them
            __builtin_amdgcn_sched_barrier(0);
            f32x16 z[KB];
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[ob][r] = bias_l[l * H + 32 * ob + mfma32_row(r, hf)];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        z[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img(L::fw(l), (ob * KB + kb) * 2 + s), hb[kb][s], z[ob], 0, 0, 0);
            }
#pragma unroll
            for (int ob = 0; ob < KB; ++ob) {
                unsigned gw[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32v2 g, d;
                    gelu_fast_pair2(f32v2{z[ob][r], z[ob][r + 1]}, g, d);
                    z[ob][r] = g[0];
                    z[ob][r + 1] = g[1];
                    gw[r >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]));
                }
                gp_store(l, ob, gw);
                const Packed16 pz = pack16(z[ob]);
                store_packed_rows(mine + L::h(l) + ob * TILE_BYTES, pz, l31, hf);
                frags_of(pz, hb[ob][0], hb[ob][1]);
            }
        }
        GNO_STAMP(1)   // MLP recompute
        // ---- last layer transposed: K'[e][c] (lane = channel, reg = edge) ------------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        f32x16 kp;
        {
            const float blv = bias_l[NH * H + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) kp[r] = blv;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    kp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb[kb][s], img(L::fw(NH), kb * 2 + s), kp, 0, 0, 0);
        }
        // ---- m' = g*k' -> grad_f (segmented sum over the source-sorted tile); dk' = g*f ----------------------------------
        if (base + 32 > E) {   // wave-uniform, last tile only: edges past E contribute nothing
            const int nvalid = (int)(E - base);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mfma32_row(r, 0) + 4 * hf >= nvalid) gv[r] = 0.f;
        }
        f32x16 dkp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dkp[r] = gv[r] * fv[r];
        if constexpr (STAMP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic: expose the gather wait here
        GNO_STAMP(2)   // last layer, wait for the gathered rows
        {
            // the rows of the tile are runs of equal source ids; a run ends where the edge is the last of its row or of the
            // tile.  The run structure is wave-uniform (ballot masks, readlane), the sums are per-lane register adds in edge
            // order -- the order of the LDS walk this replaces, so results are bit-identical to it.  m'[e][channel l31] of
            // the 16 edges the other half-wave holds arrive by permlane32_swap, eight edges at a time.
            const int64_t pos = base + l31;
            const bool ok = idv >= 0;
            const unsigned mfirst = (unsigned)__ballot(ok && (pos == (int64_t)rbv || l31 == 0));
            const unsigned mlast = (unsigned)__ballot(ok && (pos == (int64_t)rev - 1 || l31 == 31 || pos + 1 >= E));
            const unsigned mol = (unsigned)__ballot(ok && (int64_t)rbv < base);          // row open to the left
            const unsigned mor = (unsigned)__ballot(ok && (int64_t)rev > base + 32);     // row open to the right
            const int64_t tile = base >> 5;
            // running sums, restarted (by a uniform select, no branch) at the first edge of every row, parked in LDS per
            // edge; then one short uniform loop over the row ENDS picks the finished sums up and stores them
            float* runs = reinterpret_cast<float*>(mine + L::dz(0));   // [32 e][32 c] fp32: dz_0 is written later
            float run = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float lo[4], hi[4];   // edges 8g + i (half 0) and 8g + 4 + i (half 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned u = __builtin_bit_cast(unsigned, gv[4 * g + i] * kp[4 * g + i]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    lo[i] = __builtin_bit_cast(float, (unsigned)sw[0]);
                    hi[i] = __builtin_bit_cast(float, (unsigned)sw[1]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = 8 * g + k;
                    const float v = k < 4 ? lo[k] : hi[k - 4];
                    run = ((mfirst >> e) & 1u) ? v : run + v;
                    runs[e * C + l31] = run;
                }
            }
            wave_lds_fence();
            GNO_STAMP(9)   // (diagnostic split) running sums
            unsigned m = mlast;
            while (m) {   // wave-uniform
                const int e = __builtin_ctz(m);
                m &= m - 1;
                const int q = __builtin_amdgcn_readlane(idv, e);
                const float sum = runs[e * C + l31];
                if (hf == 0) {   // buffer stores: row offset in a scalar register, lane offset 4 * l31 -- no address registers
                    const bool ol = (mol >> e) & 1u, orr = (mor >> e) & 1u;
                    if (!ol && !orr) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rgf, l31 * 4, q * (C * 4), 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rpart, l31 * 4, (int)((tile * 2 + (ol ? 0 : 1)) * (C * 4)), 0);
                }
            }
            wave_lds_fence();
        }
        GNO_STAMP(3)   // segmented sums: pick-up loop
        fetch_pos();   // next tile's coordinates: its ids were requested at the top of this tile
        store_acc_rows(dkT, dkp, l31, hf);          // dk tile [c][e]
        wave_lds_fence();
        // ---- data gradients, top down: dh_l[k][e] = sum_j W_{l+1}[j][k] dz_{l+1}[j][e] ; dz_l = dh_l * gelu'(z_l) ---------
        __builtin_amdgcn_sched_barrier(0);
        f32x16 dz[KB];
        {
            const bf16x8 dk0 = frag_cols(dkT, lane, 0), dk1 = frag_cols(dkT, lane, 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img(L::bw(NH), kb * 2 + 0), dk0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img(L::bw(NH), kb * 2 + 1), dk1, acc, 0, 0, 0);
                unsigned gw[8];
                gp_load(NH - 1, kb, gw);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    dz[kb][r] = mul_f16lo(acc[r], gw[r >> 1]);
                    dz[kb][r + 1] = mul_f16hi(acc[r + 1], gw[r >> 1]);
                }
            }
        }
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 dzb[KB][2];
#pragma unroll
            for (int jb = 0; jb < KB; ++jb) {
                const Packed16 pz = pack16(dz[jb]);
                store_packed_rows(mine + L::dz(l) + jb * TILE_BYTES, pz, l31, hf);
                frags_of(pz, dzb[jb][0], dzb[jb][1]);
            }
            f32x16 dn[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dn[kb][r] = 0.f;
#pragma unroll
                for (int jb = 0; jb < KB; ++jb)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        dn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img(L::bw(l), (kb * KB + jb) * 2 + s), dzb[jb][s], dn[kb], 0, 0, 0);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                unsigned gw[8];
                gp_load(l - 1, kb, gw);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    dz[kb][r] = mul_f16lo(dn[kb][r], gw[r >> 1]);
                    dz[kb][r + 1] = mul_f16hi(dn[kb][r + 1], gw[r >> 1]);
                }
            }
        }
#pragma unroll
        for (int jb = 0; jb < KB; ++jb) store_acc_rows(mine + L::dz(0) + jb * TILE_BYTES, dz[jb], l31, hf);
        GNO_STAMP(4)   // data gradients
        __syncthreads();
        GNO_STAMP(5)   // barrier before the weight-gradient phase
        // ---- weight gradients of the four tiles: dW_l[j][k] += sum_e dz_l[j][e] h_{l-1}[k][e] ------------------------------
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const char* wb = wave_base(t);
#pragma unroll
            for (int l = 1; l < NH; ++l) {
                const char* zt = wb + L::dz(l) + wjb * TILE_BYTES;
                const char* ht = wb + L::h(l - 1) + wkb * TILE_BYTES;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 a = frag_cols_nat(zt, lane, s);
                    dWh[l - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_cols_nat(ht, lane, s), dWh[l - 1], 0, 0, 0);
                    if (wkb == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(l), bacc, 0, 0, 0);
                }
            }
            if (wave < 2) {
                // dW_L[c][k] += sum_e dk[c][e] h_{NH-1}[k][e] (k-block = wave); db_L rides on wave 0
                const char* dt = wb + L::dk;
                const char* ht = wb + L::h(NH - 1) + wave * TILE_BYTES;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 a = frag_rows(dt, l31, hf, s);
                    dWx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_cols_nat(ht, lane, s), dWx, 0, 0, 0);
                    if (wave == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(NH), bacc, 0, 0, 0);
                }
            } else {
                // dW_0[j][k] += sum_e dz_0[j][e] in[k][e] (k = 6 is the ones row -> db_0); j-block = wave - 2
                const char* zt = wb + L::dz(0) + (wave - 2) * TILE_BYTES;
                const char* it = wb + L::in;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 b = frag_rows(it, l31 & 7, hf, s);
                    if (l31 >= 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = 0;
                    }
                    dWx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_nat(zt, lane, s), b, dWx, 0, 0, 0);
                }
            }
        }
        GNO_STAMP(6)   // weight gradients
        __syncthreads();
        GNO_STAMP(7)   // barrier after
    }
    if constexpr (STAMP) {
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < 12; ++i) stamps[((int64_t)blockIdx.x * 4 + wave) * 12 + i] = tacc[i];
    }

    // ---- workgroup partial (reduced over workgroups in fixed order by k_reduce_params) ---------------------------------------
    float* wp = wpart + (int64_t)blockIdx.x * PL::total;
    if (wave < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[PL::w_off(NH) + mfma32_row(r, hf) * H + 32 * wave + l31] = dWx[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * (wave - 2) + mfma32_row(r, hf);
            if (l31 < IN0) wp[PL::w_off(0) + j * IN0 + l31] = dWx[r];
            if (l31 == 6) wp[PL::b_off(0) + j] = dWx[r];
        }
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            wp[PL::w_off(l) + (32 * wjb + mfma32_row(r, hf)) * H + 32 * wkb + l31] = dWh[l - 1][r];
        if (wkb == 0 && l31 == l) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[PL::b_off(l) + 32 * wjb + mfma32_row(r, hf)] = bacc[r];
        }
    }
    if (wave == 0 && l31 == NH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[PL::b_off(NH) + mfma32_row(r, hf)] = bacc[r];
    }
}

template <int NH>
int launch_bwd2(const void* images, const float* w0t, const MlpPtrs& p, const float* y_pos, const float* x_pos,
                const float* f_y, const float* gs, const int* src_s, const int* dst_s, const int* rowptr_src, int64_t E,
                float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    constexpr int lds = Lds2<NH>::total;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = k_gno_bwd2_bf16<NH, false>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)k_gno_bwd2_bf16<NH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gaot_set_error("gno_bwd2_bf16: cannot set dynamic LDS %d: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    static const bool stamp = getenv("GAOT_GNO_STAMPS") != nullptr;
    if (stamp) {   // diagnostic only: synchronises and prints the per-phase shares of the wave time
        unsigned long long* d = nullptr;
        const size_t n = (size_t)grid * 4 * 12;
        if (hipMalloc(&d, n * 8) != hipSuccess) return GAOT_ERR_LAUNCH;
        GAOT_KLAUNCH((k_gno_bwd2_bf16<NH, true>), dim3(grid), dim3(256), lds, st, (const uint4*)images, w0t, p, y_pos, x_pos, f_y,
                     gs, src_s, dst_s, rowptr_src, E, grad_f, part, wpart, d);
        unsigned long long* h = (unsigned long long*)malloc(n * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
        double tot[12] = {0}, all = 0;
        for (size_t i = 0; i < n; ++i) { tot[i % 12] += (double)h[i]; all += (double)h[i]; }
        const char* nm[10] = {"gather-issue", "mlp-recompute", "last-layer+gather-wait", "segment-pickup", "data-grads", "barrier-1", "weight-grads", "barrier-2", "tile-top", "running-sums"};
        fprintf(stderr, "[gno_bwd2<%d> stamps] cycles per wave %.0f:", NH, all / (grid * 4.0));
        for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * tot[i] / all);
        fprintf(stderr, "\n");
        free(h);
        (void)hipFree(d);
        return GAOT_OK;
    }
    GAOT_KLAUNCH(kern, dim3(grid), dim3(256), lds, st, (const uint4*)images, w0t, p, y_pos, x_pos, f_y, gs, src_s, dst_s,
                 rowptr_src, E, grad_f, part, wpart, (unsigned long long*)nullptr);
    return GAOT_OK;
}

}  // namespace

// images: the fragment images of k_prep_bwd_images (gno_bf16.hip), fw[1..NH-1] | fw[NH] | bw[1..NH-1] | bw[NH], contiguous
int gaot_gno_bwd2_bf16_launch(int n_hidden, const void* images, const float* w0t, const float* const* w, const float* const* b,
                              const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                              const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                              int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    MlpPtrs p;
    for (int l = 0; l <= n_hidden; ++l) { p.w[l] = w[l]; p.b[l] = b[l]; }
    switch (n_hidden) {
        case 1: return launch_bwd2<1>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 2: return launch_bwd2<2>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 3: return launch_bwd2<3>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
    }
    gaot_set_error("gaot_gno_bwd (bf16, v2): unsupported n_hidden %d", n_hidden);
    return GAOT_ERR_UNSUPPORTED;
}
