"""ctypes binding of libgaot3d_hip.so (C ABI in include/gaot3d_hip.h).

The library is built in-tree (``gaot_3d_amd/lib/libgaot3d_hip.so``) by ``__graft_entry__.build()`` /
``make -C gaot_3d_amd/csrc``.  There is NO fallback: if the library is missing or a call fails the
product raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAOT_LIB: another build of the same library (kernel A/B experiments: tools/); default = the in-tree build
LIB_PATH = os.environ.get("GAOT_LIB") or os.path.join(_HERE, "lib", "libgaot3d_hip.so")

MAX_MLP_LAYERS = 5


class GaotError(RuntimeError):
    pass


class MlpT(C.Structure):
    _fields_ = [("n_hidden", C.c_int), ("hidden", C.c_int), ("channels", C.c_int),
                ("weight", C.c_void_p * MAX_MLP_LAYERS), ("bias", C.c_void_p * MAX_MLP_LAYERS)]


class MlpGradT(C.Structure):
    _fields_ = [("weight", C.c_void_p * MAX_MLP_LAYERS), ("bias", C.c_void_p * MAX_MLP_LAYERS)]


_lib = None

_p, _i, _i64, _sz, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol include/gaot3d_hip.h declares
SIGNATURES = {
    "gaot_abi_version": (_i, []),
    "gaot_last_error": (C.c_char_p, []),
    "gaot_launch_count": (_i64, [_i]),
    "gaot_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "gaot_csr_build": (_i, [_p, _i, _i64, _i, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "gaot_gno_fwd_workspace_bytes": (_sz, [_i64, _i]),
    "gaot_gno_fwd": (_i, [C.POINTER(MlpT), _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _i, _p, _sz, _p]),
    "gaot_gno_bwd_workspace_bytes": (_sz, [C.POINTER(MlpT), _i64, _i64]),
    "gaot_gno_bwd": (_i, [C.POINTER(MlpT), _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p,
                          C.POINTER(MlpGradT), _i, _p, _sz, _p]),
    "gaot_geoembed_stats_workspace_bytes": (_sz, []),
    "gaot_geoembed_moments": (_i, [_p, _p, _p, _p, _i64, _p, _p]),
    "gaot_geoembed_from_moments": (_i, [_p, _i64, _p, _p, _sz, _p]),
    "gaot_geoembed_raw": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "gaot_geoembed_finalize": (_i, [_p, _i64, _p, _i64, _p, _sz, _p]),
    "gaot_gemm_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gaot_gemm": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _p, _i, _p, _i64, _p, _i, _p, _sz, _p]),
    "gaot_gemm_ex": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _p, _i, _p, _i64, _p, _i, _p,
                          _sz, _p]),
    "gaot_attn_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _i, _p]),
    "gaot_attn_bwd": (_i, [_p] * 10 + [_i64] * 8 + [_i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _i, _i, _p]),
    "gaot_attn_bwd_fused_f32_scratch_bytes": (_i64, [_i, _i, _i]),
    "gaot_attn_bwd_fused_f32": (_i, [_p] * 10 + [_i64] * 8 + [_i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _i, _p, C.c_size_t, _p]),
    "gaot_attn_dropout_mask": (_i, [_p, _f, _i, _i, _i, _p, _p]),
    "gaot_dropout_seed_next": (_i, [_p, C.c_uint64, _p, _p]),
    "gaot_dropout_seed_block": (_i, [_p, C.c_uint64, _i, _p, _p]),
    "gaot_attn_bf16_image_bytes": (_sz, [_i, _i, _i, _i]),
    "gaot_rope_table": (_i, [_p, _i, _i, _p, _p]),
    "gaot_qkv_image": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p, _f, _p]),
    "gaot_qkv_image_packed": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _p, _f, _i, _p]),
    "gaot_pack_heads": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p, _i, _i, _i, _p]),
    "gaot_attn_bwd_bf16_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "gaot_attn_bwd_bf16_fused_eligible": (_i, [_i, _i, _i, _i]),
    "gaot_attn_fwd_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _p]),
    "gaot_attn_bwd_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _i, _p]),
    "gaot_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p]),
    "gaot_rmsnorm_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "gaot_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _sz, _p]),
    "gaot_rmsnorm_bwd2": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _sz, _p]),
    "gaot_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "gaot_colsum": (_i, [_p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "gaot_gemm_ex_partials": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _p, _sz, _p, _p, _p]),
    "gaot_rmsnorm_bwd_parts": (_i64, [_i64]),
    "gaot_colsum_parts": (_i64, [_i64]),
    "gaot_reduce_multi": (_i, [_p, _i, _p]),
    "gaot_rope": (_i, [_p, _i64, _i64, _i, _i, _i, _i, _p, _i, _p]),
    "gaot_swiglu_fwd": (_i, [_p, _p, _i64, _i, _p]),
    "gaot_swiglu_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "gaot_adamw_step": (_i, [_p, _i, _p, _p, _d, _d, _d, _d, _p]),
    "gaot_mlp2_fwd": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "gaot_mlp2_bwd_workspace_bytes": (_sz, [_i, _i]),
    "gaot_mlp2_bwd": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gaot_knn_grid": (_i, [_p, _i64, _p, _p, _i, _p, _p]),
    "gaot_radius_grid_count": (_i, [_p, _i64, _p, _p, _f, _i, _p, _p]),
    "gaot_radius_grid_fill": (_i, [_p, _i64, _p, _p, _f, _i, _p, _p, _p, _p]),
    "gaot_knn_brute": (_i, [_p, _i64, _p, _i64, _i, _p, _p]),
    "gaot_radius_brute_count": (_i, [_p, _i64, _p, _i64, _f, _i, _p, _p]),
    "gaot_radius_brute_fill": (_i, [_p, _i64, _p, _i64, _f, _i, _p, _p, _p, _p]),
    "gaot_exclusive_scan_workspace_bytes": (_sz, [_i64]),
    "gaot_exclusive_scan_i32": (_i, [_p, _i64, _p, _p, _sz, _p]),
    "gaot_segment_cap_flags": (_i, [_p, _p, _i64, _i, _p, _p]),
    "gaot_unique_pair_flags": (_i, [_p, _p, _i64, _p, _p]),
    "gaot_random_keep_flags": (_i, [_p, _i64, _d, _p, _p]),
    "gaot_dropout": (_i, [_p, _p, _d, _i64, _p, _p]),
    "gaot_segment_random_cap_flags": (_i, [_p, _p, _p, _i64, _i, _p, _p]),
    "gaot_compact_pairs": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p]),
    "gaot_cast_bf16": (_i, [_p, _p, _i64, _p]),
    "gaot_cast_bf16_multi": (_i, [_p, _i, _p]),
    "gaot_cast_bf16_transpose_multi": (_i, [_p, _p, _p, _i, _p]),
    "gaot_swiglu_fwd_bf16": (_i, [_p, _p, _i64, _i, _p]),
    "gaot_ffn_w13_swiglu": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "gaot_ffn_w2_bwd_swiglu": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "gaot_swiglu_bwd_bf16": (_i, [_p, _p, _p, _i64, _i, _p]),
    "gaot_ffn_packed_bytes": (_i64, [_i, _i]),
    "gaot_ffn_pack": (_i, [_p, _p, _i, _p, _i, _p]),
    "gaot_ffn_pack_multi": (_i, [_p, _i, _i, _i, _p]),
    "gaot_ffn_fwd": (_i, [_p, _p, _p, _i64, _p, _p, _p, _i64, _i, _p]),
    "gaot_norm_ffn_fwd": (_i, [_p, _i64, _p, _f, _p, _p, _p, _p, _i64, _i, _p]),
    "gaot_block_packed_bytes": (_i64, [_i]),
    "gaot_block_pack_multi": (_i, [_p, _i, _i, _p]),
    "gaot_block_tail_fwd": (_i, [_p, _i64, _p, _i64, _p, _f, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "gaot_qkv_packed_bytes": (_i64, [_i64, _i]),
    "gaot_qkv_pack_multi": (_i, [_p, _i, _i64, _i, _p]),
    "gaot_norm_qkv_image": (_i, [_p, _i64, _p, _f, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _f, _p]),
    "gaot_skip_packed_bytes": (_i64, []),
    "gaot_skip_pack_multi": (_i, [_p, _i, _p]),
    "gaot_cat_norm_qkv_image": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _f, _p]),
    "gaot_norm_bwd_parts": (_i64, [_i64]),
    "gaot_ffn_bwd_norm": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "gaot_qkv_bwd_norm": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "gaot_qkv_bwd_norm_cat": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i64, _p]),
    "gaot_oproj_bwd_image": (_i, [_p, _p, _p, _i, _p, _p, _i64, _i, _p]),
    "gaot_ffn_bwd_dag": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "gaot_ffn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _p]),
    "gaot_act_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "gaot_act_fwd": (_i, [_p, _p, _i64, _i, _p]),
    "gaot_axpy": (_i, [_p, _p, _f, _p, _i64, _i64, _p]),
    "gaot_stream_copy": (_i, [_p, _p, _i64, _p]),
    "gaot_stream_copy_ex": (_i, [_p, _p, _i64, _i, _p]),
    "gaot_patchify": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gaot_mse_workspace_bytes": (_sz, []),
    "gaot_mse_fwd": (_i, [_p, _p, _i64, _p, _p, _sz, _p]),
    "gaot_mse_bwd": (_i, [_p, _p, _i64, _p, _p, _p]),
    "gaot_gather_rows": (_i, [_p, _i64, _p, _i64, _i, _p, _i64, _p]),
    "gaot_segment_reduce": (_i, [_p, _i64, _p, _p, _i64, _i, _i, _p, _p, _p]),
    "gaot_segment_reduce_bwd": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i, _p, _p]),
    "gaot_segment_softmax_fwd": (_i, [_p, _p, _i64, _p, _p]),
    "gaot_segment_softmax_bwd": (_i, [_p, _p, _p, _i64, _p, _p]),
    "gaot_edge_coords": (_i, [_p, _p, _p, _p, _i64, _i, _p, _i64, _p]),
    "gaot_mul": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "gaot_mul_rowsum": (_i, [_p, _p, _i64, _i, _p, _p]),
    "gaot_affine_cols": (_i, [_p, _p, _p, _i64, _i, _p, _p]),
    "gaot_scale_mix_fwd": (_i, [_p, _i, _p, _p, _p, _i64, _i, _p]),
    "gaot_scale_mix_bwd": (_i, [_p, _i, _p, _p, _p, _p, _i64, _i, _p]),
}


def load():
    """Load the shared library (once).  Raises GaotError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GaotError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C gaot_3d_amd/csrc` (there is no CPU/PyTorch fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gaot_abi_version() != 11:
        raise GaotError("libgaot3d_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gaot_last_error()
        raise GaotError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
