// Backward of the fused GNO integral transform on the bf16 matrix cores, third design: TWO waves per SIMD (precision 1 of
// gaot_gno_bwd; reference semantics: integral_transform.py:146-171 + LinearChannelMLP mlp.py:327-335, see gno.hip).
//
// The second design (k_gno_bwd2_bf16: 4 waves x 32-edge tiles, 512 registers and 27 KB of LDS tiles per wave) ran one
// wave per SIMD: nothing covered the MFMA -> VALU dependencies, the LDS round trips of the activations or the gather
// latency of that wave, and its vector unit was busy 49 % of the time.  Here a wave owns a 16-EDGE tile and the MLP
// runs on v_mfma_f32_16x16x32_bf16 (layer 0: v_mfma_f32_16x16x4_f32, exact), which halves every per-wave quantity --
// activations 16 registers per layer, 13 KB of LDS tiles -- so that EIGHT waves share one copy of the operand images
// (40 KB for three hidden layers) and each SIMD always has a second wave to issue from.  Everything else is as before:
//   * the MLP's operand fragments (recompute and transposed data-gradient forms) live in LDS for the whole launch, one
//     conflict-free ds_read_b128 per MFMA operand; the D tile of a layer is, register for register, the B operand of
//     the next (the image side walks k in the permuted order 32s + 16(i>>2) + 4(lane>>4) + (i&3));
//   * gathered rows are requested at the top of the tile and consumed after the MLP recompute;
//   * h_l / dz_l are rounded to bf16 once and stored as [edge][feature] tiles with 8-byte stores; every weight-gradient
//     operand is a hardware-transposed read (ds_read_b64_tr_b16) of those tiles;
//   * all weight-gradient products of the eight tiles of an iteration run in one phase between two workgroup barriers
//     (v_mfma_f32_32x32x16_bf16 with K = the 16 edges of a tile), output tiles split over the waves; bias gradients
//     ride on the same A fragments through a one-hot selector operand;
//   * the segmented sum of grad f runs over the source-sorted 16-edge tile; rows that straddle tiles go through the
//     tile-ordered fix-up (k_segment_fixup with 16-edge tiles): no atomics, bit-reproducible.
#include <stdlib.h>

#include "gno_common.h"
#include "tile32.h"

namespace {

using namespace gno;

typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));

// feature that element i of k-step s stands for in a lane of group g = lane >> 4 (accumulator-as-operand order)
__device__ __host__ __forceinline__ constexpr int kperm(int s, int g, int i) { return 32 * s + 16 * (i >> 2) + 4 * g + (i & 3); }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32v2 t = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16v2));
}
// two D tiles (feature blocks 2s, 2s+1) -> the B / A fragment of k-step s
__device__ __forceinline__ bf16x8 frag_of(const f32x4& lo, const f32x4& hi) {
    return __builtin_bit_cast(bf16x8, make_uint4(pk_bf16(lo[0], lo[1]), pk_bf16(lo[2], lo[3]), pk_bf16(hi[0], hi[1]), pk_bf16(hi[2], hi[3])));
}
// the same words as 8-byte pieces of the [16 e][64 f] LDS tile (two half tiles of [16][32], tile32 layout): lane (edge n,
// group g) holds features 16 mb + 4 g + 0..3 of block mb
// The [16 e][32 f] half tiles of h_l / dz_l are WRITTEN with 8-byte stores (ds_write_b64: groups of 16 consecutive lanes = the 16 edge
// rows of one group g, banks taken modulo 32 dwords) and READ only through ds_read_b64_tr_b16.  In the plain tile32 layout a row is
// 64 B = 16 dwords, so rows n and n + 2 of a group land on the same banks: every such store ran 2-way conflicted (round 5 counters:
// SQ_LDS_BANK_CONFLICT 26 M / 34 M of 47.7 M / 67.7 M LDS cycles for NH = 2 / 3; ~26 stores of ~4 extra cycles per tile).  Here the
// 8-byte half of a chunk is swapped for rows with bit 1 set -- 16 rows x 8 B cover all 32 banks -- and the transposed read applies
// the same swap (a bijection of the half-wave's addresses: still conflict free under its modulo-64 banking).
__device__ __forceinline__ int tile_off8(int row, int chunk, int half) { return tile_off(row, chunk) + 8 * (half ^ ((row >> 1) & 1)); }
__device__ __forceinline__ void store_frag_rows(char* tile, const bf16x8& f, int s, int n, int g) {
    const uint4 w = __builtin_bit_cast(uint4, f);
    char* half = tile + s * 1024;   // blocks 2s, 2s+1 = features 32 s .. 32 s + 31 = half tile s
    *reinterpret_cast<uint2*>(half + tile_off8(n, (g >> 1), g & 1)) = make_uint2(w.x, w.y);
    *reinterpret_cast<uint2*>(half + tile_off8(n, 2 + (g >> 1), g & 1)) = make_uint2(w.z, w.w);
}
// contract over tile ROWS (edges 0..15), lane = column: element i <-> row 8 hf + i  (the natural k order of a row read)
__device__ __forceinline__ bf16x8 frag_cols16(const char* tile, int lane) {
    const int i = lane & 15, grp = (lane >> 4) & 1, hf = lane >> 5;
    const int col = 16 * grp + 4 * (i & 3);
    const int r0 = 8 * hf + (i >> 2), r1 = r0 + 4;
    const char* p0 = tile + tile_off8(r0, col >> 3, (col & 7) >> 2);      // tiles written by store_frag_rows
    const char* p1 = tile + tile_off8(r1, col >> 3, (col & 7) >> 2);
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}
// B operand of a 16x16x32 product from a [32 rows][16 cols] tile: lane (column n, group g), element i <-> row 8 g + i
__device__ __forceinline__ bf16x8 frag_rows8(const char* tile, int n, int g) {
    const int col = 4 * (n & 3);
    const int r0 = 8 * g + (n >> 2), r1 = r0 + 4;
    const char* p0 = tile + tile_off(r0, col >> 3) + ((col & 7) << 1);
    const char* p1 = tile + tile_off(r1, col >> 3) + ((col & 7) << 1);
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}

// a * (f16 half of g): an fma with a +0 addend cannot be folded to a multiply (signed zeros), so the compiler selects
// v_fma_mix_f32, which reads the f16 half directly
__device__ __forceinline__ float mul_f16lo(float a, unsigned g) { return __builtin_fmaf(a, (float)__builtin_bit_cast(h16x2, g)[0], 0.0f); }
__device__ __forceinline__ float mul_f16hi(float a, unsigned g) { return __builtin_fmaf(a, (float)__builtin_bit_cast(h16x2, g)[1], 0.0f); }

template <int NH>
struct Lds3 {
    static constexpr int H = 64, C = 32, WAVES = 8;
    static constexpr int per_hidden = 8 * 1024;   // 4 row blocks x 2 k-steps fragments of 64 lanes x 16 B
    static constexpr int per_last = 4 * 1024;
    static constexpr int img_bytes = 2 * ((NH - 1) * per_hidden + per_last);   // fw[1..NH-1], fw[NH], bw[1..NH-1], bw[NH]
    static constexpr int fw(int l) { return (l - 1) * per_hidden; }            // l = 1..NH (NH = last)
    static constexpr int bw(int l) { return (NH - 1) * per_hidden + per_last + (l - 1) * per_hidden; }
    // four hidden layers: 56 KB of images + eight waves' tiles (8 x 18.5 KB) do not fit 160 KB -- the fragments are then read
    // from the image buffer in global memory (1 KB per wave instruction, coalesced, L2 resident) instead of LDS
    static constexpr bool GIMG = NH >= 4;
    // float [8][W0S] (rows 6, 7 zero); row stride 80, not 64: a lane group of the layer-0 operand read (ds_read_b32, groups of 32
    // lanes, banks modulo 32) holds rows g and g + 1 -- 64 apart they share every bank (2-way), 80 apart none
    static constexpr int W0S = 80;
    static constexpr int w0t = GIMG ? 0 : img_bytes;
    static constexpr int bias = w0t + 8 * W0S * 4;              // float NH*64 + 32
    static constexpr int tiles = (bias + (NH * H + C) * 4 + 15) & ~15;
    static constexpr int h(int l) { return l * 2048; }          // per-wave offsets: [16 e][64 f] bf16
    static constexpr int dz(int l) { return NH * 2048 + l * 2048; }
    static constexpr int dk = 2 * NH * 2048;                    // [32 c][16 e] bf16 (tile32 layout, columns 0..15)
    static constexpr int in = dk + 2048;                        // [8 k][16 e] bf16
    static constexpr int per_wave = in + 512;
    static constexpr int total = tiles + WAVES * per_wave;
};

template <int NH>
struct ParamLayout3 {  // flat per-workgroup partial layout, state_dict order (same as gno.hip / gno_bf16.hip)
    static constexpr int H = 64, C = 32;
    static constexpr int w_off(int l) { return l == 0 ? 0 : (H * IN0 + H) + (l - 1) * (H * H + H); }
    static constexpr int b_off(int l) { return w_off(l) + (l == 0 ? H * IN0 : (l == NH ? C * H : H * H)); }
    static constexpr int total = (H * IN0 + H) + (NH - 1) * (H * H + H) + (C * H + C);
};

// fragment images of the kernel below (one 16-byte piece per lane and fragment)
__global__ void k_prep_bwd3_images(MlpPtrs mlp, int nh, bf16_t* base) {
    constexpr int H = 64;
    const int per_hidden = 8 * 64 * 8, per_last = 4 * 64 * 8;   // elements
    const int total = (nh - 1) * 2 * per_hidden + 2 * per_last;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int o = idx;
        float v;
        if (o < (nh - 1) * per_hidden) {                       // fw[l]: A = W_l rows 16 mb + m, k-step s
            const int l = 1 + o / per_hidden; o %= per_hidden;
            const int i = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, mb = fr >> 1;
            v = mlp.w[l][(16 * mb + (ln & 15)) * H + kperm(s, ln >> 4, i)];
        } else if ((o -= (nh - 1) * per_hidden) < per_last) {  // fw[nh]: B of the transposed last layer, channels 16 nb + n
            const int i = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, nb = fr >> 1;
            v = mlp.w[nh][(16 * nb + (ln & 15)) * H + kperm(s, ln >> 4, i)];
        } else if ((o -= per_last) < (nh - 1) * per_hidden) {  // bw[l]: A = W_l^T rows k = 16 mb + m, elements j
            const int l = 1 + o / per_hidden; o %= per_hidden;
            const int i = o & 7, ln = (o >> 3) & 63, fr = o >> 9, s = fr & 1, mb = fr >> 1;
            v = mlp.w[l][kperm(s, ln >> 4, i) * H + 16 * mb + (ln & 15)];
        } else {                                               // bw[nh]: A = W_L^T rows k = 16 mb + m, elements c = 8 g + i
            o -= (nh - 1) * per_hidden;
            const int i = o & 7, ln = (o >> 3) & 63, mb = o >> 9;
            v = mlp.w[nh][(8 * (ln >> 4) + i) * H + 16 * mb + (ln & 15)];
        }
        base[idx] = f2bf(v);
    }
}

template <int NH>
__global__ __launch_bounds__(512, 1) void k_gno_bwd3_bf16(
    const uint4* __restrict__ images, const float* __restrict__ w0t_g, MlpPtrs mlp, const float* __restrict__ y_pos,
    const float* __restrict__ x_pos, const float* __restrict__ f_y, const float* __restrict__ gs,
    const int* __restrict__ src_s, const int* __restrict__ dst_s, const int* __restrict__ rowptr_src, int64_t E,
    float* __restrict__ grad_f, float* __restrict__ part, float* __restrict__ wpart) {
    constexpr int C = 32, H = 64;
    using L = Lds3<NH>;
    using PL = ParamLayout3<NH>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, l31 = lane & 31, hf = lane >> 5;

    // ---- resident operands: fragment images (k_prep_bwd3_images), layer-0 weight, biases ------------------------------
    if constexpr (!L::GIMG)
        for (int i = threadIdx.x; i < L::img_bytes / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = images[i];
    {
        float* w0 = reinterpret_cast<float*>(lds + L::w0t);
        for (int i = threadIdx.x; i < 8 * H; i += 512) w0[(i >> 6) * L::W0S + (i & 63)] = (i < IN0 * H) ? w0t_g[i] : 0.f;
        float* bl = reinterpret_cast<float*>(lds + L::bias);
#pragma unroll
        for (int l = 0; l < NH; ++l)
            for (int i = threadIdx.x; i < H; i += 512) bl[l * H + i] = mlp.b[l][i];
        for (int i = threadIdx.x; i < C; i += 512) bl[NH * H + i] = mlp.b[NH][i];
    }
    __syncthreads();
    const float* w0 = reinterpret_cast<const float*>(lds + L::w0t);
    const float* bias_l = reinterpret_cast<const float*>(lds + L::bias);
    auto img = [&](int off_bytes, int frag) {
        if constexpr (L::GIMG) return reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(images) + off_bytes)[frag * 64 + lane];
        else return reinterpret_cast<const bf16x8*>(lds + off_bytes)[frag * 64 + lane];
    };
    auto wave_base = [&](int w) { return lds + L::tiles + w * L::per_wave; };
    char* mine = wave_base(wave);
    char* dkT = mine + L::dk;
    char* inT = mine + L::in;

    // ---- weight-gradient jobs of this wave (wave-uniform):
    //   H(l, jb, kb): dW_l[32 jb.., 32 kb..] over all 8 tiles (+ db_l[32 jb..] on the kb = 0 wave)
    //   X(x): x = 0, 1: dW_L[:, 32 x..] (+ db_L on x = 0);  x = 2, 3: dW_0[32 (x-2).., :] (column 6 = the ones row = db_0)
    // NH = 3: every wave one H job and one X job over half of the tiles; NH = 2: waves 0-3 H, waves 4-7 X, all tiles;
    // NH = 1: X over half of the tiles.  Halves are added in the epilogue (waves 4-7 into 0-3), a fixed order.
    // NH = 4 (sixteen jobs, two per wave over all tiles): every wave H(l = 1 + (wave >> 2), jb, kb); waves 0-3 also H(3, jb, kb),
    // waves 4-7 the X jobs.  The bias gradients of all of a wave's jobs share `bacc`: job (layer l) lands in column l.
    const bool has_h = (NH >= 3) || (NH == 2 && wave < 4);
    const bool has_h2 = (NH == 4) && wave < 4;
    const bool has_x = (NH == 1 || NH == 3) || wave >= 4;
    const int hl = (NH >= 3) ? 1 + (wave >> 2) : 1, hjb = (wave >> 1) & 1, hkb = wave & 1;
    const int xj = wave & 3;
    const bool x_all = NH == 2 || NH == 4;
    const int xt0 = x_all ? 0 : 4 * (wave >> 2), xt1 = x_all ? 8 : xt0 + 4;
    f32x16 acc_h, acc_x, bacc, acc_h2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_h[r] = 0.f; acc_x[r] = 0.f; bacc[r] = 0.f; acc_h2[r] = 0.f; }
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
    const int64_t n_tiles = (E + 15) / 16;
    // ids and endpoint coordinates of a tile are two DEPENDENT global round trips: fetched one iteration ahead
    int s_nx = 0, q_nx = 0;
    bool v_nx = false;
    float bin_nx[2] = {0.f, 0.f};
    auto fetch_ids = [&](int64_t tb_) {
        const int64_t e = (tb_ + wave) * 16 + n;
        v_nx = tb_ < n_tiles && e < E;
        s_nx = v_nx ? src_s[e] : 0;
        q_nx = v_nx ? dst_s[e] : 0;
    };
    auto fetch_pos = [&]() {   // lane (edge n, group g): input rows g and 4 + g of [y(3), x(3), 1, 0]
        const float* ys = y_pos + (int64_t)s_nx * 3;
        const float* xq = x_pos + (int64_t)q_nx * 3;
        bin_nx[0] = g < 3 ? ys[g] : xq[0];
        bin_nx[1] = g < 2 ? xq[1 + g] : 0.f;
    };
    fetch_ids((int64_t)blockIdx.x * 8);
    fetch_pos();
    for (int64_t tb = (int64_t)blockIdx.x * 8; tb < n_tiles; tb += (int64_t)gridDim.x * 8) {
        const int64_t base = (tb + wave) * 16;
        // compiler-only memory barrier: without it the loop-invariant LDS reads of the operand images are hoisted out of the
        // tile loop into (and beyond) the whole register file
        asm volatile("" ::: "memory");
        float bin0, bin1;
        int idv, rbv = 0, rev = 0;   // per edge (lane n, all groups alike): source row or -1, and its [rb, re) edge range
        int s_raw, q_raw;            // endpoints as fetched (0 for edges past E)
        {
            const bool valid = v_nx;
            const int s = s_nx, q = q_nx;
            bin0 = bin_nx[0]; bin1 = bin_nx[1];
            idv = valid ? s : -1;
            s_raw = s;
            q_raw = q;
            // input tile [k][e] bf16 (rows 0..5 = coordinates, row 6 = ones -> db_0, row 7 = 0).  Consumes the coordinates
            // requested at the end of the previous iteration BEFORE anything new is requested: vmcnt retires in order, so a wait
            // for them placed behind fresh requests would sit out those requests' whole round trip.
            *reinterpret_cast<bf16_t*>(inT + tile_off(g, n >> 3) + ((n & 7) << 1)) = f2bf(bin0);
            *reinterpret_cast<bf16_t*>(inT + tile_off(4 + g, n >> 3) + ((n & 7) << 1)) =
                g < 2 ? f2bf(bin1) : (g == 2 ? (bf16_t)0x3F80 : (bf16_t)0);
            asm volatile("" ::: "memory");
            // UNCONDITIONAL loads (s = 0 for an edge past E: a valid row), selected afterwards: inside `if (valid)` the loads, their
            // s_waitcnt vmcnt(0) and the merge sit in one block at the top of the iteration
            rbv = rowptr_src[s];
            rev = rowptr_src[s + 1];
            rbv = valid ? rbv : 0;
            rev = valid ? rev : 0;
            fetch_ids(tb + (int64_t)gridDim.x * 8);
        }
        // ---- gather f[src] / g[dst] rows now (lane = channel 16 nb + n, reg = edge 4 g + r): they are needed after the MLP
        //      recompute.  Unconditional loads (edges past E carry endpoint 0, a valid row); zeroed at their use.
        float fv[2][4], gv[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int el = 4 * g + r;
            const int s_ = __builtin_amdgcn_ds_bpermute(4 * el, s_raw), q_ = __builtin_amdgcn_ds_bpermute(4 * el, q_raw);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                gv[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, q_ * (C * 4) + (16 * nb + n) * 4, 0, 0));
                fv[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, s_ * (C * 4) + (16 * nb + n) * 4, 0, 0));
            }
        }
        // ---- recompute the MLP: h_l goes to LDS as bf16 [edge][feature]; gelu'(z_l) as packed f16 pairs: layer 0 in
        //      registers, the upper layers parked in the LDS slot their own dz_l tile overwrites right after consuming it
        unsigned gp0[8];
        auto gp_store = [&](int l, const unsigned (&w)[8]) {
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gp0[j] = w[j];
                return;
            }
            char* p = mine + L::dz(l) + lane * 16;
            *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(p + 1024) = make_uint4(w[4], w[5], w[6], w[7]);
        };
        auto gp_load = [&](int l, unsigned (&w)[8]) {
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = gp0[j];
                return;
            }
            const char* p = mine + L::dz(l) + lane * 16;
            const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 1024);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        bf16x8 hb[2];
        // activation of one layer: z (4 row blocks) -> gelu in place, gelu' packed, fragments + LDS tile of h_l
        auto activate = [&](int l, f32x4 (&z)[4]) {
            unsigned gw[8];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32v2 gl, d;
                    gelu_e2_pair2(f32v2{z[mb][2 * p], z[mb][2 * p + 1]}, gl, d);
                    z[mb][2 * p] = gl[0];
                    z[mb][2 * p + 1] = gl[1];
                    gw[2 * mb + p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]));
                }
            gp_store(l, gw);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                hb[s] = frag_of(z[2 * s], z[2 * s + 1]);
                store_frag_rows(mine + L::h(l), hb[s], s, n, g);
            }
        };
        {
            f32x4 z[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                z[mb] = *reinterpret_cast<const f32x4*>(bias_l + 16 * mb + 4 * g);
                z[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[g * L::W0S + 16 * mb + n], bin0, z[mb], 0, 0, 0);
                z[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[(4 + g) * L::W0S + 16 * mb + n], bin1, z[mb], 0, 0, 0);
            }
            activate(0, z);
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            __builtin_amdgcn_sched_barrier(0);
            f32x4 z[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                z[mb] = *reinterpret_cast<const f32x4*>(bias_l + l * H + 16 * mb + 4 * g);
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    z[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img(L::fw(l), mb * 2 + s), hb[s], z[mb], 0, 0, 0);
            }
            activate(l, z);
        }
        // ---- last layer transposed: K'[e][c] (lane = channel 16 nb + n, reg = edge 4 g + r) --------------------------------
        __builtin_amdgcn_sched_barrier(0);
        f32x4 kp[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float blv = bias_l[NH * H + 16 * nb + n];
            kp[nb] = f32x4{blv, blv, blv, blv};
#pragma unroll
            for (int s = 0; s < 2; ++s)
                kp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb[s], img(L::fw(NH), nb * 2 + s), kp[nb], 0, 0, 0);
        }
        // ---- m' = g*k' -> grad_f (segmented sum over the source-sorted tile); dk' = g*f ----------------------------------
        if (base + 16 > E) {   // wave-uniform, last tiles only: edges past E contribute nothing
            const int nvalid = (int)(E - base);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r >= nvalid) { gv[0][r] = 0.f; gv[1][r] = 0.f; }
        }
        {
            // the rows of the tile are runs of equal source ids; a run ends where the edge is the last of its row or of the
            // tile.  m'[e][c] goes through a [16 e][32 c] fp32 scratch (the dz_0 slot, written later): every lane (channel
            // l31, half hf) turns the 8 edges of its half into running sums, restarted at the first edge of every row;
            // a short wave-uniform loop over the row ENDS picks the finished sums up (a row that began in the first half and
            // ends in the second adds the first half's last running sum) and stores them.
            float* runs = reinterpret_cast<float*>(mine + L::dz(0));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) runs[(4 * g + r) * C + 16 * nb + n] = gv[nb][r] * kp[nb][r];
            const int64_t pos = base + n;
            const bool ok = idv >= 0;
            const unsigned mfirst = (unsigned)__ballot(ok && (pos == (int64_t)rbv || n == 0)) & 0xffffu;
            const unsigned mlast = (unsigned)__ballot(ok && (pos == (int64_t)rev - 1 || n == 15 || pos + 1 >= E)) & 0xffffu;
            const unsigned mol = (unsigned)__ballot(ok && (int64_t)rbv < base) & 0xffffu;          // row open to the left
            const unsigned mor = (unsigned)__ballot(ok && (int64_t)rev > base + 16) & 0xffffu;     // row open to the right
            const int64_t tile = base >> 4;
            wave_lds_fence();
            float run = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = 8 * hf + k;
                const float v = runs[e * C + l31];
                run = ((mfirst >> e) & 1u) ? v : run + v;
                runs[e * C + l31] = run;
            }
            wave_lds_fence();
            unsigned m = mlast;
            while (m) {   // wave-uniform
                const int e = __builtin_ctz(m);
                m &= m - 1;
                const int q = __builtin_amdgcn_readlane(idv, e);
                float sum = runs[e * C + l31];
                if (e >= 8 && ((mfirst >> 8) & ((2u << (e - 8)) - 1u)) == 0u) sum = runs[7 * C + l31] + sum;
                if (hf == 0) {   // buffer stores: row offset in a scalar register, lane offset 4 * l31
                    const bool ol = (mol >> e) & 1u, orr = (mor >> e) & 1u;
                    if (!ol && !orr) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rgf, l31 * 4, q * (C * 4), 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rpart, l31 * 4, (int)((tile * 2 + (ol ? 0 : 1)) * (C * 4)), 0);
                }
            }
            wave_lds_fence();
        }
        fetch_pos();   // next tile's coordinates: its ids were requested at the top of this tile
        // dk tile [c][e]: row 16 nb + n, columns 4 g .. 4 g + 3
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            *reinterpret_cast<uint2*>(dkT + tile_off(16 * nb + n, g >> 1) + 8 * (g & 1)) =
                make_uint2(pk_bf16(gv[nb][0] * fv[nb][0], gv[nb][1] * fv[nb][1]), pk_bf16(gv[nb][2] * fv[nb][2], gv[nb][3] * fv[nb][3]));
        wave_lds_fence();
        // ---- data gradients, top down: dh_l[k][e] = sum_j W_{l+1}[j][k] dz_{l+1}[j][e] ; dz_l = dh_l * gelu'(z_l) ---------
        __builtin_amdgcn_sched_barrier(0);
        f32x4 dz[4];
        {
            const bf16x8 dkf = frag_rows8(dkT, n, g);
            unsigned gw[8];
            gp_load(NH - 1, gw);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img(L::bw(NH), mb), dkf, acc, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    dz[mb][2 * p] = mul_f16lo(acc[2 * p], gw[2 * mb + p]);
                    dz[mb][2 * p + 1] = mul_f16hi(acc[2 * p + 1], gw[2 * mb + p]);
                }
            }
        }
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 dzb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dzb[s] = frag_of(dz[2 * s], dz[2 * s + 1]);
                store_frag_rows(mine + L::dz(l), dzb[s], s, n, g);
            }
            unsigned gw[8];
            gp_load(l - 1, gw);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 dn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    dn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img(L::bw(l), mb * 2 + s), dzb[s], dn, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    dz[mb][2 * p] = mul_f16lo(dn[2 * p], gw[2 * mb + p]);
                    dz[mb][2 * p + 1] = mul_f16hi(dn[2 * p + 1], gw[2 * mb + p]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) store_frag_rows(mine + L::dz(0), frag_of(dz[2 * s], dz[2 * s + 1]), s, n, g);
        __syncthreads();
        // ---- weight gradients of the eight tiles: dW_l[j][k] += sum_e dz_l[j][e] h_{l-1}[k][e]  (K = the 16 edges) ---------
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            const char* wb = wave_base(t);
            if (NH >= 2 && has_h) {
                const bf16x8 a = frag_cols16(wb + L::dz(hl) + hjb * 1024, lane);
                acc_h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_cols16(wb + L::h(hl - 1) + hkb * 1024, lane), acc_h, 0, 0, 0);
                if (hkb == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(hl), bacc, 0, 0, 0);
            }
            if (NH == 4 && has_h2) {
                const bf16x8 a = frag_cols16(wb + L::dz(NH - 1) + hjb * 1024, lane);
                acc_h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_cols16(wb + L::h(NH - 2) + hkb * 1024, lane), acc_h2, 0, 0, 0);
                if (hkb == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(NH - 1), bacc, 0, 0, 0);
            }
            if (has_x && t >= xt0 && t < xt1) {
                if (xj < 2) {
                    // dW_L[c][k] += sum_e dk[c][e] h_{NH-1}[k][e] (k-block = xj); db_L rides on xj = 0
                    const bf16x8 a = frag_rows(wb + L::dk, l31, hf, 0);
                    acc_x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_cols16(wb + L::h(NH - 1) + xj * 1024, lane), acc_x, 0, 0, 0);
                    if (xj == 0) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, sel(NH), bacc, 0, 0, 0);
                } else {
                    // dW_0[j][k] += sum_e dz_0[j][e] in[k][e] (k = 6 is the ones row -> db_0); j-block = xj - 2
                    bf16x8 b = frag_rows(wb + L::in, l31 & 7, hf, 0);
                    if (l31 >= 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = 0;
                    }
                    acc_x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols16(wb + L::dz(0) + (xj - 2) * 1024, lane), b, acc_x, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- halves of the X jobs (and of db_L): waves 4-7 hand theirs to waves 0-3 through LDS, added in that order -----------
    if (NH == 1 || NH == 3) {
        float* xs = reinterpret_cast<float*>(mine);   // 16 x 64 floats per wave (4 KB of its tile area)
        float* bs = reinterpret_cast<float*>(lds);    // bacc of wave 4: the image area is free now (>= 8 KB)
        if (wave >= 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[r * 64 + lane] = acc_x[r];
            if (wave == 4)
#pragma unroll
                for (int r = 0; r < 16; ++r) bs[r * 64 + lane] = bacc[r];
        }
        __syncthreads();
        if (wave < 4) {
            const float* ps = reinterpret_cast<const float*>(wave_base(wave + 4));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_x[r] += ps[r * 64 + lane];
            if (wave == 0 && l31 == NH)
#pragma unroll
                for (int r = 0; r < 16; ++r) bacc[r] += bs[r * 64 + lane];
        }
    }

    // ---- workgroup partial (reduced over workgroups in fixed order by k_reduce_params) ---------------------------------------
    float* wp = wpart + (int64_t)blockIdx.x * PL::total;
    if (has_x && (x_all || wave < 4)) {
        if (xj < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[PL::w_off(NH) + mfma32_row(r, hf) * H + 32 * xj + l31] = acc_x[r];
            if (xj == 0 && l31 == NH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) wp[PL::b_off(NH) + mfma32_row(r, hf)] = bacc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * (xj - 2) + mfma32_row(r, hf);
                if (l31 < IN0) wp[PL::w_off(0) + j * IN0 + l31] = acc_x[r];
                if (l31 == 6) wp[PL::b_off(0) + j] = acc_x[r];
            }
        }
    }
    if (NH >= 2 && has_h) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            wp[PL::w_off(hl) + (32 * hjb + mfma32_row(r, hf)) * H + 32 * hkb + l31] = acc_h[r];
        if (hkb == 0 && l31 == hl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[PL::b_off(hl) + 32 * hjb + mfma32_row(r, hf)] = bacc[r];
        }
    }
    if (NH == 4 && has_h2) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            wp[PL::w_off(NH - 1) + (32 * hjb + mfma32_row(r, hf)) * H + 32 * hkb + l31] = acc_h2[r];
        if (hkb == 0 && l31 == NH - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[PL::b_off(NH - 1) + 32 * hjb + mfma32_row(r, hf)] = bacc[r];
        }
    }
}

template <int NH>
int launch_bwd3(const void* images, const float* w0t, const MlpPtrs& p, const float* y_pos, const float* x_pos,
                const float* f_y, const float* gs, const int* src_s, const int* dst_s, const int* rowptr_src, int64_t E,
                float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    constexpr int lds = Lds3<NH>::total;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = k_gno_bwd3_bf16<NH>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gaot_set_error("gno_bwd3_bf16: cannot set dynamic LDS %d: %s", lds, hipGetErrorString(e));
            return GAOT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    GAOT_KLAUNCH(kern, dim3(grid), dim3(512), lds, st, (const uint4*)images, w0t, p, y_pos, x_pos, f_y, gs, src_s, dst_s,
                 rowptr_src, E, grad_f, part, wpart);
    return GAOT_OK;
}

}  // namespace

// images: scratch of gaot_gno_bwd_bf16_image_bytes() (gno_bf16.hip); w0t = fp32 [6][64] transposed first-layer weight; part: two
// 32-channel slots per 16-EDGE tile (k_segment_fixup<32, 4>); wpart: one flat parameter-gradient partial per workgroup
int gaot_gno_bwd3_bf16_launch(int n_hidden, void* images, const float* w0t, const float* const* w, const float* const* b,
                              const float* y_pos, const float* x_pos, const float* f_y, const float* gs,
                              const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* rowptr_src,
                              int64_t num_edges, float* grad_f, float* part, float* wpart, int grid, hipStream_t st) {
    if (n_hidden < 1 || n_hidden > 4) {      // before anything indexes w / b / MlpPtrs with it
        gaot_set_error("gaot_gno_bwd (bf16): unsupported n_hidden %d", n_hidden);
        return GAOT_ERR_UNSUPPORTED;
    }
    MlpPtrs p;
    for (int l = 0; l <= n_hidden; ++l) { p.w[l] = w[l]; p.b[l] = b[l]; }
    GAOT_KLAUNCH(k_prep_bwd3_images, dim3(32), dim3(256), 0, st, p, n_hidden, (bf16_t*)images);
    switch (n_hidden) {
        case 1: return launch_bwd3<1>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 2: return launch_bwd3<2>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        case 3: return launch_bwd3<3>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
        default: return launch_bwd3<4>(images, w0t, p, y_pos, x_pos, f_y, gs, src_sorted, dst_sorted, rowptr_src, num_edges, grad_f, part, wpart, grid, st);
    }
}
