// 32 x 32 bf16 LDS tiles shared by the attention and the fused-MLP kernels: one image serves the operand that contracts
// over the tile's columns (ds_read_b128 along a row) and the one that contracts over its rows (ds_read_b64_tr_b16,
// hardware transpose), both bank-conflict free.  Anonymous namespace: one copy per including translation unit.
#pragma once
#include "common.h"

namespace {

typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// ---- LDS tile: [32 rows][32 bf16] = 2 KB, 16-B chunk c of row r lives at chunk c ^ ((r>>2)&3) ----------
constexpr int TILE_BYTES = 32 * 64;
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// operand that contracts over head_dim: lane (row = l31, half hf), k-step s -> 8 bf16 = cols 16s+8hf..+7
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int l31, int hf, int s) {
    return *reinterpret_cast<const bf16x8*>(tile + tile_off(l31, 2 * s + hf));
}
// operand that contracts over tile ROWS (transposed use): lane (col = l31, half hf), k-step s ->
// element j = tile[row 16s + 8(j>>2) + 4hf + (j&3)][col]  (the k order of an accumulator-as-operand)
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int lane, int s) {
    const int i = lane & 15, grp = (lane >> 4) & 1, hf = lane >> 5;
    const int col = 16 * grp + 4 * (i & 3);          // this lane ADDRESSES row (i>>2), cols col..col+3 of the block
    const int r0 = 16 * s + 4 * hf + (i >> 2);
    const int r1 = r0 + 8;
    const char* p0 = tile + tile_off(r0, col >> 3) + ((col & 7) << 1);
    const char* p1 = tile + tile_off(r1, col >> 3) + ((col & 7) << 1);
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p1));
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}
// fp32 accumulator tile (16 regs) -> two bf16 B-operand fragments (k-steps 0,1)
__device__ __forceinline__ void acc_to_frags(const f32x16& p, bf16x8& f0, bf16x8& f1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f0[j] = (short)f2bf(p[j]);
        f1[j] = (short)f2bf(p[8 + j]);
    }
}


}  // namespace
