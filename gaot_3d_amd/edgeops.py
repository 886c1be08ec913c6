"""Autograd operators over csrc/edgeops.hip: the general (unfused) per-edge path for the GNO variants the fused
kernels do not cover -- IntegralTransform ``transform_type`` "nonlinear" / "nonlinear_kernelonly", segment-softmax
attention weights (reference src/model/layers/integral_transform.py:68-78, 126-171), kernel MLPs of other shapes,
PointNet GeometricEmbedding (src/model/layers/geoembed.py:184-222).

All per-edge tensors are in the dst-sorted edge order of ``ops.build_graph`` (``g.by_dst``): ``src = g.by_dst.other``,
``dst = g.by_dst.key``; a query's edges are the contiguous rows ``rowptr[q] : rowptr[q+1]``.  The per-edge MLP runs
through ``functional.linear`` (GEMM kernels with bias / GELU / ReLU epilogues).  HIP only: CPU tensors raise."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch.autograd import Function

from . import _lib
from ._lib import check
from .ops import BipartiteGraph, _ptr, _req, _stream

Tensor = torch.Tensor
SUM, MEAN, MAX, MIN = 0, 1, 2, 3


def _off(t: Tensor, elems: int) -> C.c_void_p:
    return C.c_void_p(t.data_ptr() + 4 * elems)


def src_to_dst_map(g: BipartiteGraph) -> Tensor:
    """position in the src-sorted order -> position in the dst-sorted order (int32 [E]); cached on the graph"""
    m = g.__dict__.get("_src2dst")
    if m is None:
        e = g.by_dst.num_edges
        inv = torch.empty(e, dtype=torch.int32, device=g.by_dst.perm.device)
        inv[g.by_dst.perm.long()] = torch.arange(e, dtype=torch.int32, device=inv.device)
        m = inv[g.by_src.perm.long()].contiguous()
        g.__dict__["_src2dst"] = m
    return m


# ---- raw calls --------------------------------------------------------------------------------------------------
def gather_rows(table: Tensor, idx: Tensor, out: Optional[Tensor] = None, col0: int = 0) -> Tensor:
    lib = _lib.load()
    table = _req(table, torch.float32, "table")
    e, c = idx.shape[0], table.shape[1]
    if out is None:
        out = torch.empty(e, c, dtype=torch.float32, device=table.device)
    check(lib.gaot_gather_rows(_ptr(table), c, _ptr(idx), e, c, _off(out, col0), out.shape[1], _stream()), "gaot_gather_rows")
    return out


def segment_reduce(vals: Tensor, rowptr: Tensor, pos_map: Optional[Tensor], num_rows: int, mode: int, col0: int = 0,
                   channels: Optional[int] = None, want_argmax: bool = False):
    lib = _lib.load()
    vals = _req(vals, torch.float32, "vals")
    c = vals.shape[1] - col0 if channels is None else channels
    out = torch.empty(num_rows, c, dtype=torch.float32, device=vals.device)
    arg = torch.empty(num_rows, c, dtype=torch.int32, device=vals.device) if (mode in (MAX, MIN) and want_argmax) else None
    check(lib.gaot_segment_reduce(_off(vals, col0), vals.shape[1], _ptr(rowptr), _ptr(pos_map), num_rows, c, mode, _ptr(out),
                                  _ptr(arg), _stream()), "gaot_segment_reduce")
    return (out, arg) if want_argmax else out


def mul(a: Tensor, b: Tensor, row_scalar: bool = False) -> Tensor:
    lib = _lib.load()
    a = _req(a, torch.float32, "a")
    b = _req(b, torch.float32, "b")
    out = torch.empty_like(a)
    rows = a.shape[0]
    c = a.numel() // max(rows, 1) if rows else 1
    check(lib.gaot_mul(_ptr(a), _ptr(b), rows, max(c, 1), int(row_scalar), _ptr(out), _stream()), "gaot_mul")
    return out


def mul_rowsum(a: Tensor, b: Tensor) -> Tensor:
    lib = _lib.load()
    a = _req(a, torch.float32, "a")
    b = _req(b, torch.float32, "b")
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    check(lib.gaot_mul_rowsum(_ptr(a), _ptr(b), a.shape[0], a.shape[1], _ptr(out), _stream()), "gaot_mul_rowsum")
    return out


def edge_coords(y_pos: Tensor, x_pos: Tensor, g: BipartiteGraph, mode: int, out: Optional[Tensor] = None) -> Tensor:
    """mode 0: [y[src], x[dst]] -> [E,6]; 1: y[src] - x[dst] -> [E,3]; 2: cosine(x[dst], y[src]) -> [E]"""
    lib = _lib.load()
    y_pos = _req(y_pos, torch.float32, "y_pos")
    x_pos = _req(x_pos, torch.float32, "x_pos")
    if y_pos.shape[1] != 3 or x_pos.shape[1] != 3:
        raise _lib.GaotError("edge_coords: coordinates must be [*, 3]")
    e = g.by_dst.num_edges
    if out is None:
        out = torch.empty((e, 6) if mode == 0 else (e, 3) if mode == 1 else (e,), dtype=torch.float32, device=x_pos.device)
    ld = out.shape[1] if out.dim() == 2 else 1
    check(lib.gaot_edge_coords(_ptr(y_pos), _ptr(x_pos), _ptr(g.by_dst.other), _ptr(g.by_dst.key), e, mode, _ptr(out), ld,
                               _stream()), "gaot_edge_coords")
    return out


# ---- autograd ---------------------------------------------------------------------------------------------------
class GatherFn(Function):
    """table[src] (side 0) or table[dst] (side 1) per edge; backward = fixed-order segment sum over that endpoint"""

    @staticmethod
    def forward(ctx, table: Tensor, g: BipartiteGraph, side: int):
        ctx.g, ctx.side, ctx.rows = g, side, table.shape[0]
        return gather_rows(table, g.by_dst.other if side == 0 else g.by_dst.key)

    @staticmethod
    def backward(ctx, d: Tensor):
        g = ctx.g
        d = d if d.is_contiguous() else d.contiguous()
        if ctx.side == 0:
            return segment_reduce(d, g.by_src.rowptr, src_to_dst_map(g), ctx.rows, SUM), None, None
        return segment_reduce(d, g.by_dst.rowptr, None, ctx.rows, SUM), None, None


class EdgeInputFn(Function):
    """agg = cat([y_pos[src], x_pos[dst], f_y[src] (optional)], -1)  (integral_transform.py:146-152); grad -> f_y only"""

    @staticmethod
    def forward(ctx, y_pos: Tensor, x_pos: Tensor, f_y: Optional[Tensor], g: BipartiteGraph):
        e = g.by_dst.num_edges
        c = 0 if f_y is None else f_y.shape[1]
        out = torch.empty(e, 6 + c, dtype=torch.float32, device=x_pos.device)
        edge_coords(y_pos, x_pos, g, 0, out)
        if f_y is not None:
            gather_rows(f_y, g.by_dst.other, out, col0=6)
        ctx.g, ctx.c, ctx.rows = g, c, (0 if f_y is None else f_y.shape[0])
        return out

    @staticmethod
    def backward(ctx, d: Tensor):
        if ctx.c == 0:
            return None, None, None, None
        g = ctx.g
        d = d if d.is_contiguous() else d.contiguous()
        df = segment_reduce(d, g.by_src.rowptr, src_to_dst_map(g), ctx.rows, SUM, col0=6, channels=ctx.c)
        return None, None, df, None


class SegmentReduceFn(Function):
    """scatter(vals, dst, reduce) over the queries (scatter_native.py:4-54): sum / mean / max / min, empty rows -> 0"""

    @staticmethod
    def forward(ctx, vals: Tensor, g: BipartiteGraph, mode: int):
        ctx.g, ctx.mode, ctx.shape = g, mode, tuple(vals.shape)
        if mode in (MAX, MIN):
            out, arg = segment_reduce(vals, g.by_dst.rowptr, None, g.num_dst, mode, want_argmax=True)
            ctx.save_for_backward(arg)
            return out
        return segment_reduce(vals, g.by_dst.rowptr, None, g.num_dst, mode)

    @staticmethod
    def backward(ctx, d: Tensor):
        lib = _lib.load()
        g = ctx.g
        d = _req(d, torch.float32, "d_out")
        e, c = ctx.shape
        dv = torch.empty(e, c, dtype=torch.float32, device=d.device)
        arg = ctx.saved_tensors[0] if ctx.mode in (MAX, MIN) else None
        check(lib.gaot_segment_reduce_bwd(_ptr(d), _ptr(g.by_dst.key), _ptr(g.by_dst.rowptr), _ptr(arg), g.num_dst, e, c,
                                          ctx.mode, _ptr(dv), _stream()), "gaot_segment_reduce_bwd")
        return dv, None, None


class ScatterFn(Function):
    """torch_scatter.scatter / scatter_native (reference src/model/layers/utils/scatter_native.py:4-54) on an UNSORTED
    index: one stable radix sort of the index (csr_build) gives every output row its contributions in input order, then
    the fixed-order segment reduction -- no atomics, bit-reproducible; empty rows -> 0."""

    @staticmethod
    def forward(ctx, src2: Tensor, index: Tensor, dim_size: int, mode: int):
        from . import ops
        e, c = src2.shape
        idx = index.to(torch.int32) if index.dtype != torch.int32 else index
        se = ops.csr_build(torch.stack([idx, idx]), 1, dim_size)          # perm: sorted position -> input position
        ctx.mode, ctx.shape, ctx.rows = mode, (e, c), dim_size
        if mode in (MAX, MIN):
            out, arg = segment_reduce(src2, se.rowptr, se.perm, dim_size, mode, want_argmax=True)
            ctx.save_for_backward(idx, se.rowptr, arg)
            return out
        ctx.save_for_backward(idx, se.rowptr)
        return segment_reduce(src2, se.rowptr, se.perm, dim_size, mode)

    @staticmethod
    def backward(ctx, d: Tensor):
        lib = _lib.load()
        d = _req(d, torch.float32, "d_out")
        e, c = ctx.shape
        saved = ctx.saved_tensors
        arg = saved[2] if ctx.mode in (MAX, MIN) else None
        dv = torch.empty(e, c, dtype=torch.float32, device=d.device)
        check(lib.gaot_segment_reduce_bwd(_ptr(d), _ptr(saved[0]), _ptr(saved[1]), _ptr(arg), ctx.rows, e, c, ctx.mode,
                                          _ptr(dv), _stream()), "gaot_segment_reduce_bwd")
        return dv, None, None, None


_REDUCE = {"sum": SUM, "add": SUM, "mean": MEAN, "max": MAX, "amax": MAX, "min": MIN, "amin": MIN}


def scatter(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None, dim_size: Optional[int] = None,
            reduce: str = "sum") -> Tensor:
    """Drop-in for the reference's ``scatter`` (scatter_native.py:4-54 / torch_scatter.scatter as the reference calls it:
    dim=0, 1-D index over the leading axis).  Same argument meaning and errors: dim != 0 raises NotImplementedError,
    an unknown ``reduce`` ValueError; ``out`` (if given) is overwritten like the reference's ``out.fill_(0)`` + scatter;
    dim_size=None takes index.max()+1 (a host read, as in the reference)."""
    if dim != 0:
        raise NotImplementedError("Native scatter fallback only supports dim=0")
    if reduce not in _REDUCE:
        raise ValueError(f"Unsupported reduce operation '{reduce}' in native scatter")
    if not src.is_cuda:
        raise _lib.GaotError(f"scatter: expected tensors on the GPU (the HIP path has no CPU fallback), got {src.device}")
    if index.dim() != 1 or index.shape[0] != src.shape[0]:
        raise ValueError("scatter: index must be 1-D over the leading axis of src")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    tail = tuple(src.shape[1:])
    c = 1
    for t in tail:
        c *= t
    if src.shape[0] == 0 or dim_size == 0 or c == 0:
        res = torch.zeros((dim_size,) + tail, dtype=src.dtype, device=src.device)
    else:
        res = ScatterFn.apply(src.reshape(src.shape[0], c).to(torch.float32), index, dim_size, _REDUCE[reduce])
        res = res.view((dim_size,) + tail).to(src.dtype)
    if out is not None:
        out.copy_(res)
        return out
    return res


class SegmentSoftmaxFn(Function):
    """softmax of per-edge scores over each query's edges (integral_transform.py:68-78)"""

    @staticmethod
    def forward(ctx, scores: Tensor, g: BipartiteGraph):
        lib = _lib.load()
        s = _req(scores, torch.float32, "scores")
        w = torch.empty_like(s)
        check(lib.gaot_segment_softmax_fwd(_ptr(s), _ptr(g.by_dst.rowptr), g.num_dst, _ptr(w), _stream()),
              "gaot_segment_softmax_fwd")
        ctx.g = g
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, dw: Tensor):
        lib = _lib.load()
        (w,) = ctx.saved_tensors
        dw = _req(dw, torch.float32, "dw")
        ds = torch.empty_like(w)
        check(lib.gaot_segment_softmax_bwd(_ptr(w), _ptr(dw), _ptr(ctx.g.by_dst.rowptr), ctx.g.num_dst, _ptr(ds), _stream()),
              "gaot_segment_softmax_bwd")
        return ds, None


class MulFn(Function):
    """a .* b, same shapes"""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ctx.save_for_backward(a, b)
        return mul(a, b)

    @staticmethod
    def backward(ctx, d: Tensor):
        a, b = ctx.saved_tensors
        d = d if d.is_contiguous() else d.contiguous()
        return (mul(d, b) if ctx.needs_input_grad[0] else None), (mul(d, a) if ctx.needs_input_grad[1] else None)


class RowScaleFn(Function):
    """a[rows, C] * w[rows, None]"""

    @staticmethod
    def forward(ctx, a: Tensor, w: Tensor):
        ctx.save_for_backward(a, w)
        return mul(a, w, row_scalar=True)

    @staticmethod
    def backward(ctx, d: Tensor):
        a, w = ctx.saved_tensors
        d = d if d.is_contiguous() else d.contiguous()
        return (mul(d, w, row_scalar=True) if ctx.needs_input_grad[0] else None), \
               (mul_rowsum(d, a) if ctx.needs_input_grad[1] else None)


class RowDotFn(Function):
    """sum_c a[row, c] * b[row, c] * scale"""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, scale: float):
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        s = mul_rowsum(a, b)
        return s * scale if scale != 1.0 else s

    @staticmethod
    def backward(ctx, d: Tensor):
        a, b = ctx.saved_tensors
        d = (d * ctx.scale).contiguous()
        return mul(b, d, row_scalar=True), mul(a, d, row_scalar=True), None


def affine_cols(x: Tensor, scale_m1: Tensor, bias: Optional[Tensor]) -> Tensor:
    lib = _lib.load()
    x = _req(x, torch.float32, "x")
    scale_m1 = _req(scale_m1, torch.float32, "scale")
    bias = None if bias is None else _req(bias, torch.float32, "bias")
    out = torch.empty_like(x)
    check(lib.gaot_affine_cols(_ptr(x), _ptr(scale_m1), _ptr(bias), x.shape[0], x.shape[1], _ptr(out), _stream()), "gaot_affine_cols")
    return out


class AffineColsFn(Function):
    """x[rows, C] * (1 + s[C]) + b[C]  -- ConditionedNorm.forward for one batch element (reference mlp.py:112-128)"""

    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor, b: Tensor):
        ctx.save_for_backward(x, s)
        return affine_cols(x, s, b)

    @staticmethod
    def backward(ctx, d: Tensor):
        from . import ops
        x, s = ctx.saved_tensors
        d = d if d.is_contiguous() else d.contiguous()
        dx = affine_cols(d, s, None)
        rows, c = x.shape
        ds = ops.colsum(mul(d, x), rows, c, c)
        db = ops.colsum(d, rows, c, c)
        return dx, ds, db
