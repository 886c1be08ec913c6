"""Tensor-level wrappers over the C ABI: take torch tensors (device memory + stream plumbing only),
pass raw pointers to libgaot3d_hip.so.  Every function requires CUDA(HIP) tensors and raises otherwise."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GaotError, MlpGradT, MlpT, check

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t: Tensor, dtype, name: str) -> Tensor:
    if not t.is_cuda:
        raise GaotError(f"{name}: expected a tensor on the GPU (the HIP path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise GaotError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# CSR
# ------------------------------------------------------------------------------------------------
@dataclass
class SortedEdges:
    """Edge list sorted (stably) by one endpoint.  ``key`` = the sort endpoint, ``other`` = the other."""
    rowptr: Tensor   # int32 [num_rows+1]
    perm: Tensor     # int32 [E]
    key: Tensor      # int32 [E]
    other: Tensor    # int32 [E]
    num_rows: int

    @property
    def num_edges(self) -> int:
        return int(self.key.shape[0])


def csr_build(edge_index: Tensor, sort_row: int, num_rows: int) -> SortedEdges:
    lib = _lib.load()
    if edge_index.dtype not in (torch.int32, torch.int64):
        raise GaotError(f"edge_index must be int32 or int64, got {edge_index.dtype}")
    ei = _req(edge_index, edge_index.dtype, "edge_index")
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise GaotError(f"edge_index must be [2,E], got {tuple(ei.shape)}")
    e = int(ei.shape[1])
    dev = ei.device
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    key = torch.empty(e, dtype=torch.int32, device=dev)
    other = torch.empty(e, dtype=torch.int32, device=dev)
    nb = lib.gaot_csr_workspace_bytes(e, num_rows)
    ws = _ws(nb, dev)
    check(lib.gaot_csr_build(_ptr(ei), int(ei.dtype == torch.int64), e, sort_row, num_rows, _ptr(rowptr), _ptr(perm),
                             _ptr(key), _ptr(other), _ptr(ws), ws.numel(), _stream()), "gaot_csr_build")
    return SortedEdges(rowptr, perm, key, other, num_rows)


@dataclass
class BipartiteGraph:
    """Both orderings of one reference ``edge_index`` (row 0 = source, row 1 = query)."""
    by_dst: SortedEdges   # key = query,  other = source
    by_src: SortedEdges   # key = source, other = query
    num_src: int
    num_dst: int


def build_graph(edge_index: Tensor, num_src: int, num_dst: int) -> BipartiteGraph:
    return BipartiteGraph(csr_build(edge_index, 1, num_dst), csr_build(edge_index, 0, num_src), num_src, num_dst)


# ------------------------------------------------------------------------------------------------
# GNO integral transform
# ------------------------------------------------------------------------------------------------
def _mlp_struct(weights: Sequence[Tensor], biases: Sequence[Tensor]) -> Tuple[MlpT, List[Tensor]]:
    n = len(weights)
    if n < 2 or n > _lib.MAX_MLP_LAYERS:
        raise GaotError(f"kernel MLP must have 2..{_lib.MAX_MLP_LAYERS} linear layers, got {n}")
    keep = []
    m = MlpT()
    m.n_hidden = n - 1
    m.hidden = int(weights[0].shape[0])
    m.channels = int(weights[-1].shape[0])
    for l, (w, b) in enumerate(zip(weights, biases)):
        if w.dim() == 3:  # Conv1d(k=1) storage of mlp_type='channel'
            w = w[:, :, 0]
        w = _req(w, torch.float32, f"mlp.weight[{l}]")
        b = _req(b, torch.float32, f"mlp.bias[{l}]")
        exp_in = 6 if l == 0 else m.hidden
        exp_out = m.channels if l == n - 1 else m.hidden
        if tuple(w.shape) != (exp_out, exp_in):
            raise GaotError(f"kernel MLP layer {l}: weight shape {tuple(w.shape)} unsupported by the fused HIP path "
                            f"(needs [{exp_out},{exp_in}]: coord_dim 3, equal hidden widths)")
        keep += [w, b]
        m.weight[l] = w.data_ptr()
        m.bias[l] = b.data_ptr()
    return m, keep


def gno_forward(weights, biases, y_pos: Tensor, x_pos: Tensor, f_y: Tensor, g: BipartiteGraph) -> Tensor:
    lib = _lib.load()
    m, keep = _mlp_struct(weights, biases)
    y_pos = _req(y_pos, torch.float32, "y_pos")
    x_pos = _req(x_pos, torch.float32, "x_pos")
    f_y = _req(f_y, torch.float32, "f_y")
    if f_y.shape[1] != m.channels:
        raise GaotError(f"f_y has {f_y.shape[1]} channels, kernel MLP outputs {m.channels}")
    q = g.num_dst
    e = g.by_dst.num_edges
    out = torch.empty(q, m.channels, dtype=torch.float32, device=x_pos.device)
    ws = _ws(lib.gaot_gno_fwd_workspace_bytes(e, m.channels), x_pos.device)
    check(lib.gaot_gno_fwd(C.byref(m), _ptr(y_pos), _ptr(x_pos), _ptr(f_y), _ptr(g.by_dst.other), _ptr(g.by_dst.key),
                           _ptr(g.by_dst.rowptr), e, q, _ptr(out), _ptr(ws), ws.numel(), _stream()), "gaot_gno_fwd")
    return out


def gno_backward(weights, biases, y_pos: Tensor, x_pos: Tensor, f_y: Tensor, grad_out: Tensor, g: BipartiteGraph):
    """-> (grad_f_y, [grad_w...], [grad_b...])"""
    lib = _lib.load()
    m, keep = _mlp_struct(weights, biases)
    y_pos = _req(y_pos, torch.float32, "y_pos")
    x_pos = _req(x_pos, torch.float32, "x_pos")
    f_y = _req(f_y, torch.float32, "f_y")
    grad_out = _req(grad_out, torch.float32, "grad_out")
    e = g.by_src.num_edges
    dev = x_pos.device
    grad_f = torch.empty(g.num_src, m.channels, dtype=torch.float32, device=dev)
    gw = [torch.empty(tuple(w.shape), dtype=torch.float32, device=dev) for w in weights]
    gb = [torch.empty(tuple(b.shape), dtype=torch.float32, device=dev) for b in biases]
    gs = MlpGradT()
    for l in range(len(weights)):
        gs.weight[l] = gw[l].data_ptr()
        gs.bias[l] = gb[l].data_ptr()
    ws = _ws(lib.gaot_gno_bwd_workspace_bytes(C.byref(m), e), dev)
    check(lib.gaot_gno_bwd(C.byref(m), _ptr(y_pos), _ptr(x_pos), _ptr(f_y), _ptr(grad_out), _ptr(g.by_dst.rowptr),
                           _ptr(g.by_src.key), _ptr(g.by_src.other), _ptr(g.by_src.rowptr), e, g.num_src, g.num_dst,
                           _ptr(grad_f), C.byref(gs), _ptr(ws), ws.numel(), _stream()), "gaot_gno_bwd")
    return grad_f, gw, gb
