"""autograd.Function wrappers: forward AND backward of every hot-path operator run hand-written HIP
kernels through the C ABI (gaot_3d_amd.ops).  PyTorch only owns the device buffers and chains the
Functions.  Nothing here falls back to ATen math or to the CPU."""
from __future__ import annotations

import os
from typing import List, Optional

import torch
from torch.autograd import Function

from . import ops
from ._lib import GaotError

Tensor = torch.Tensor


def _w2d(w: Tensor) -> Tensor:
    """nn.Linear [out,in] or Conv1d(k=1) [out,in,1] storage -> contiguous [out,in] view."""
    if w.dim() == 3:
        w = w[:, :, 0]
    return w if w.is_contiguous() else w.contiguous()


# bf16 weight copies made ahead of their use by ONE multi-tensor launch (precast_weights, called by Transformer.forward
# for all of its matrices) and valid only until release_precast(): keyed by (address, shape) of the fp32 matrix
_WB_CACHE: dict = {}


def _wb_eligible(w: Tensor, prec: int) -> bool:
    n, k = w.shape
    return bool(prec == 1 and w.is_cuda and n > 64 and k > 64 and n % 8 == 0 and k % 8 == 0 and w.data_ptr() % 16 == 0)


def fused_view(ws) -> Tensor:
    """the [sum(out_i), in] matrix formed by co-located weights (see colocate)"""
    w0 = _w2d(ws[0])
    return w0.new_empty(0).set_(w0.untyped_storage(), w0.storage_offset(), (sum(_w2d(w).shape[0] for w in ws), w0.shape[1]),
                                (w0.shape[1], 1))


_WBT_CACHE: dict = {}


def precast_weights(mats, transposed=()) -> None:
    """``transposed``: matrices whose bf16 TRANSPOSE is wanted as well (the FFN's w2 and w1|w3: their input-gradient GEMMs run
    as x W^T on it, see FFNFn.backward) -- one more launch for all of them"""
    _WB_CACHE.clear()
    _WBT_CACHE.clear()
    if ops.get_precision() != "bf16":
        return
    todo = [m for m in (_w2d(m) for m in mats) if _wb_eligible(m, 1)]
    for m, c in zip(todo, ops.cast_bf16_multi(todo)):
        _WB_CACHE[(m.data_ptr(), tuple(m.shape))] = c
    if transposed and torch.is_grad_enabled():
        todo = [m for m in (_w2d(m) for m in transposed) if _wb_eligible(m, 1) and m.is_contiguous()]
        for m, c in zip(todo, ops.cast_bf16_transpose_multi(todo)):
            _WBT_CACHE[(m.data_ptr(), tuple(m.shape))] = c


# fragment-ordered images of FFN weights for the fused FFN kernels (csrc/ffn_fused.hip), made by ONE launch for all blocks of a
# Transformer (prepack_ffn) and valid until release_precast(): keyed by (address, shape) of the co-located fp32 [w1; w3]
_FFN_PACK_CACHE: dict = {}
_WO_PACKED: dict = {}      # key of a block image that carries the o_proj fragment image -> address of that o_proj weight
_FFN_FUSED = os.environ.get("GAOT_FFN_FUSED", "1") != "0"
_BLOCK_TAIL = os.environ.get("GAOT_BLOCK_TAIL", "1") != "0"    # o_proj + residual + ffn_norm + FFN + residual in one forward launch (A/B switch)
_OPROJ_BWD_IMAGE = os.environ.get("GAOT_OPROJ_BWD_IMAGE", "1") != "0"   # BlockTailFn.backward hands dO over as the flash backward's image (A/B switch)
_NORM_FFN = os.environ.get("GAOT_NORM_FFN", "1") != "0"        # the block's ffn_norm inside the fused FFN forward (A/B switch)
_FFN_BWD_DX = os.environ.get("GAOT_FFN_BWD_DX", "1") != "0"      # the input gradient inside the fused backward launch (A/B switch)


def _ffn_fusable(w13: Tensor, w2: Tensor) -> bool:
    f2, d = w13.shape
    return bool(_FFN_FUSED and ops.get_precision() == "bf16" and w13.is_cuda and d == 256 and f2 % 256 == 0 and tuple(w2.shape) == (256, f2 // 2)
                and w13.dtype == torch.float32 and w2.dtype == torch.float32 and w13.is_contiguous() and w2.is_contiguous())


def prepack_ffn(pairs, with_backward: bool, wos=None) -> None:
    """``pairs``: (co-located [w1; w3] view, w2) of every FFN about to run; those the fused kernels take (d_model 256, F % 128 == 0,
    bf16 mode) are packed by one launch per distinct F; ``with_backward``: the images the fused backward reads as well; ``wos``
    (with the backward images): the o_proj weight of each pair's block -- packed behind them for the block-tail kernel (BlockTailFn)"""
    _FFN_PACK_CACHE.clear()
    _WO_PACKED.clear()
    by_f: dict = {}
    pairs = list(pairs)
    wos = list(wos) if (wos is not None and with_backward and _BLOCK_TAIL) else [None] * len(pairs)
    for (w13, w2), wo in zip(pairs, wos):
        w13, w2 = _w2d(w13), _w2d(w2)
        if _ffn_fusable(w13, w2):
            if wo is not None:
                wo = _w2d(wo)
                if tuple(wo.shape) != (256, 256) or wo.dtype != torch.float32 or not wo.is_contiguous():
                    wo = None
            by_f.setdefault((w13.shape[0] // 2, wo is not None), []).append((w13, w2, wo))
    for (f, has_wo), items in by_f.items():
        if has_wo:
            outs = ops.block_pack_multi(items, f)
        else:
            outs = ops.ffn_pack_multi([(a, b) for a, b, _ in items], f, with_backward)
        for (w13, _w2, wo), packed in zip(items, outs):
            _FFN_PACK_CACHE[(w13.data_ptr(), tuple(w13.shape))] = (packed, bool(with_backward))
            if has_wo:
                _WO_PACKED[(w13.data_ptr(), tuple(w13.shape))] = wo.data_ptr()


_QKV_PACK_CACHE: dict = {}       # (address, shape) of the co-located fp32 q | k | v weight -> its fragment image (prepack_qkv)
_NORM_QKV = os.environ.get("GAOT_NORM_QKV", "1") != "0"      # attn_norm + q | k | v image in one launch (A/B switch)
_NORM_BWD_FUSED = os.environ.get("GAOT_NORM_BWD_FUSED", "1") != "0"    # RMSNorm backward in the epilogue of the product in front of it (A/B switch)


def prepack_qkv(wcats) -> None:
    """``wcats``: the co-located [q; k; v] weight views of the blocks about to run; those the fused head kernel takes (bf16 mode,
    [N, 256] with N a multiple of 256) are packed by one launch"""
    _QKV_PACK_CACHE.clear()
    if not (_NORM_QKV and ops.get_precision() == "bf16"):
        return
    by_n: dict = {}
    for w in wcats:
        w = _w2d(w)
        if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == 256 and w.shape[0] % 256 == 0:
            by_n.setdefault(w.shape[0], []).append(w)
    for _n, ws in by_n.items():
        for w, packed in zip(ws, ops.qkv_pack_multi(ws, True)):
            _QKV_PACK_CACHE[(w.data_ptr(), tuple(w.shape))] = packed


_SKIP_PACK_CACHE: dict = {}      # (address, shape) of a decoder block's fp32 skip_proj weight -> its fragment image (prepack_skip)
_CAT_BWD_DX = os.environ.get("GAOT_CAT_BWD_DX", "1") != "0"  # ... and its two input gradients inside the head's backward kernel (A/B switch)
_CAT_QKV = os.environ.get("GAOT_CAT_QKV", "1") != "0"        # the decoder block's skip projection inside the head kernel (A/B switch)


def prepack_skip(ws) -> None:
    _SKIP_PACK_CACHE.clear()
    if not (_CAT_QKV and _NORM_QKV and ops.get_precision() == "bf16"):
        return
    ok = [w for w in (_w2d(w) for w in ws) if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (256, 512)]
    for w, packed in zip(ok, ops.skip_pack_multi(ok)):
        _SKIP_PACK_CACHE[(w.data_ptr(), tuple(w.shape))] = packed


def release_precast() -> None:
    _WB_CACHE.clear()
    _WBT_CACHE.clear()
    _FFN_PACK_CACHE.clear()
    _QKV_PACK_CACHE.clear()
    _SKIP_PACK_CACHE.clear()


def _wbt(w: Tensor) -> Optional[Tensor]:
    """the bf16 transpose of a weight prepared by precast_weights, if any"""
    return _WBT_CACHE.get((w.data_ptr(), tuple(w.shape)))


def _wb(w: Tensor, precision: Optional[int]) -> Tensor:
    """the weight as the bf16 GEMMs' B operand: in bf16 mode a wide weight ([out > 64, in > 64], 16-byte rows) is rounded
    to bf16 ONCE per use instead of once per workgroup that streams it (a 64-row tile re-reads the whole matrix: at
    M = 16 384 that is 256 x); everything else stays fp32"""
    prec = (1 if ops.get_precision() == "bf16" else 0) if precision is None else precision
    if _wb_eligible(w, prec):
        c = _WB_CACHE.get((w.data_ptr(), tuple(w.shape)))
        return c if c is not None else ops.cast_bf16(w)
    return w


def _dw_gemm(a: Tensor, b: Tensor, m: int, n: int, k: int, lda: int, ldb: int, precision, params=None) -> Tensor:
    """a weight-gradient product dW[m, n] = a^T b over the rows.  Nothing in the backward chain reads it, so with a side stream
    set (``ShardedStep(overlap_dw=True)`` -> ``comm.set_side_stream``) it is issued there, beside the chain's kernels and the
    exchange steps, and joined once before the optimizer (``comm.side_join``); otherwise a plain call -- whose split-K completion,
    when the caller names the parameters the product is the gradient of (``params``) and ``ops.defer_ok`` vouches for them, is
    left to the ONE reduction launch at the end of the backward pass."""
    from . import comm
    if comm.side_stream() is None:
        return ops.gemm_dw(a, b, m, n, k, lda, ldb, precision, defer=ops.defer_ok(params))
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    comm.side_run(lambda: ops.gemm(a, b, m, n, k, lda, ldb, True, False, out=out, ldc=n, precision=precision), (a, b, out))
    return out


class LinearFn(Function):
    """y = act(x W^T + b) [+ residual].  Stands in for nn.Linear (+ F.gelu / ReLU, + the residual add that follows
    it in the Transformer block) and its autograd."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int, precision: Optional[int],
                residual: Optional[Tensor]):
        w = _w2d(weight)
        n, k = w.shape
        if x.shape[-1] != k:
            raise GaotError(f"linear: input has {x.shape[-1]} features, weight expects {k}")
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        m = x2.shape[0]
        res = None
        if residual is not None:
            res = residual.reshape(m, n)
            if not res.is_contiguous():
                res = res.contiguous()
        w = _wb(w, precision)
        if act > 3:     # outside the GEMM epilogue's set (ops.ACT): pre-activation from the GEMM, activation as its own pass
            if res is not None:
                raise GaotError("linear: a residual together with an activation outside none / gelu / relu / silu")
            z = ops.gemm(x2, w, m, n, k, k, k, False, True, bias, 0, precision=precision)
            y = ops.act_fwd(z, act)
        elif act:
            y, z = ops.gemm(x2, w, m, n, k, k, k, False, True, bias, act, residual=res, ldr=n, want_preact=True,
                            precision=precision)
        else:
            y, z = ops.gemm(x2, w, m, n, k, k, k, False, True, bias, 0, residual=res, ldr=n, precision=precision), None
        ctx.save_for_backward(x2, w, z)
        ctx.act, ctx.has_bias, ctx.precision = act, bias is not None, precision
        ctx.wshape, ctx.xshape = weight.shape, x.shape
        ctx.wparam, ctx.bparam = weight, bias     # the parameters themselves (the saved w may be a bf16 copy): ops.defer_ok
        ctx.res_shape = residual.shape if residual is not None else None
        return y.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, w, z = ctx.saved_tensors
        n, k = w.shape
        m = x2.shape[0]
        dy2 = dy.reshape(m, n)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dz = ops.act_bwd(z, dy2, ctx.act) if ctx.act else dy2
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dz, w, m, k, n, n, k, False, False, precision=ctx.precision).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            if n == 1:   # dW[0][:] = sum_m dz[m] x[m][:] -- put the wide dimension on the tile rows instead
                dw = _dw_gemm(x2, dz, k, 1, m, k, 1, ctx.precision, (ctx.wparam,)).view(ctx.wshape)
            else:
                dw = _dw_gemm(dz, x2, n, k, m, n, k, ctx.precision, (ctx.wparam,)).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dz, m, n, n, defer=ops.defer_ok((ctx.bparam,)))
        if ctx.res_shape is not None and ctx.needs_input_grad[5]:
            dres = dy2.view(ctx.res_shape)
        return dx, dw, db, None, None, dres


def linear(x, weight, bias=None, act: Optional[str] = None, precision: Optional[int] = None, residual=None):
    return LinearFn.apply(x, weight, bias, ops.ACT[act], precision, residual)


class GnoFn(Function):
    """Fused IntegralTransform (transform_type='linear', mean reduction)."""

    @staticmethod
    def forward(ctx, f_y: Tensor, y_pos: Tensor, x_pos: Tensor, graph, *params):
        nl = len(params) // 2
        ws = [_w2d(p) for p in params[0::2]]
        bs = list(params[1::2])
        f = f_y if f_y.is_contiguous() else f_y.contiguous()
        out = ops.gno_forward(ws, bs, y_pos, x_pos, f, graph)
        ctx.graph = graph
        ctx.nl = nl
        ctx.wshapes = [p.shape for p in params[0::2]]
        ctx.save_for_backward(f, y_pos, x_pos, *ws, *bs)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        saved = ctx.saved_tensors
        f, y_pos, x_pos = saved[:3]
        ws = list(saved[3:3 + ctx.nl])
        bs = list(saved[3 + ctx.nl:])
        d = dout if dout.is_contiguous() else dout.contiguous()
        gf, gw, gb = ops.gno_backward(ws, bs, y_pos, x_pos, f, d, ctx.graph)
        grads: List[Optional[Tensor]] = []
        for l in range(ctx.nl):
            grads += [gw[l].view(ctx.wshapes[l]), gb[l]]
        return (gf, None, None, None, *grads)


def _want_bf16_copy(x: Tensor) -> bool:
    return ops.get_precision() == "bf16" and x.is_cuda and x.shape[-1] % 8 == 0


def bf16_copy_of(x: Tensor, shape) -> Optional[Tensor]:
    """the bf16 image its producer (RMSNorm) attached to x, if any: the GEMM's A operand without an in-kernel conversion"""
    xb = getattr(x, "_gaot_bf16", None)
    if xb is None or xb.numel() != x.numel() or not xb.is_contiguous():
        return None
    return xb.view(shape)


class RMSNormFn(Function):
    """-> (y, bf16 copy of y or an empty tensor).  The copy is not differentiable: it is the same values, for the GEMM"""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, eps: float):
        xc = x if x.is_contiguous() else x.contiguous()
        y, rstd, yb = ops.rmsnorm_fwd(xc, weight, eps, _want_bf16_copy(xc))
        ctx.save_for_backward(xc, weight, rstd)
        yb = yb if yb is not None else torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(yb)
        ctx.set_materialize_grads(False)   # no zero-filled [rows, d] gradient for the bf16 copy
        return y, yb

    @staticmethod
    def backward(ctx, dy: Optional[Tensor], _dyb=None):
        x, w, rstd = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(x)
        d = dy if dy.is_contiguous() else dy.contiguous()
        dx, dw = ops.rmsnorm_bwd(x, w, d, rstd, defer=ops.defer_ok((w,)))
        return dx, dw, None


# Attention-dropout seed: one int64 word per device, living ON the device.  Every dropout forward takes a copy (kept for
# its backward) and advances the word with device ops, so a captured hipGraph draws a fresh mask on every replay.
# Seeded from torch.initial_seed() (torch.manual_seed controls it) on first use and whenever that seed changes.
_DROP_STATE: dict = {}
_DROP_STRIDE = -7046029254386353131   # 0x9E3779B97F4A7C15 as int64


def set_dropout_seed(seed: int, device) -> None:
    dev = torch.device(device)
    v = int(seed) & 0xFFFFFFFFFFFFFFFF
    v = v - (1 << 64) if v >= (1 << 63) else v
    _DROP_STATE[dev] = (torch.tensor([v], dtype=torch.int64, device=dev), torch.initial_seed())
    _DROP_RESERVED.pop(dev, None)


_DROP_RESERVED: dict = {}       # device -> list of one-element views of a reserved block of seed words, next first


def _drop_state(device):
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _DROP_STATE.get(dev)
    if st is None or st[1] != torch.initial_seed():
        set_dropout_seed(torch.initial_seed() * 0x2545F4914F6CDD1D + 0x632BE59BD9B4E019, dev)
        st = _DROP_STATE[dev]
    return dev, st


def reserve_dropout_seeds(device, n: int) -> None:
    """draw the seed words of the next ``n`` dropout calls on ``device`` with ONE launch (a Transformer's L attention layers: L launches
    of ~4 us otherwise); next_dropout_seed hands them out in order -- the same words the unreserved calls would read.  Words left
    over at release_dropout_seeds are skipped (the stream has advanced past them)."""
    if n <= 0:
        return
    dev, st = _drop_state(device)
    block = ops.dropout_seed_block(st[0], _DROP_STRIDE, n)
    _DROP_RESERVED[dev] = [block[i:i + 1] for i in range(n)]


def release_dropout_seeds(device) -> None:
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    _DROP_RESERVED.pop(dev, None)


def next_dropout_seed(device) -> Tensor:
    """the seed word for one dropout call (a one-element int64 device tensor); advances the device state"""
    dev, st = _drop_state(device)
    res = _DROP_RESERVED.get(dev)
    if res:
        return res.pop(0)
    return ops.dropout_seed_next(st[0], _DROP_STRIDE)


def dropout_seed_sequence(seed0: int, n: int):
    """the unsigned 64-bit words the next n dropout calls read after set_dropout_seed(seed0) (for checks)"""
    return [(int(seed0) + i * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def _all_gather_stack(t: Tensor, group, gsz: int) -> Tensor:
    from .sharding import all_gather_stack
    return all_gather_stack(t, group, gsz)


class RMSNormResFn(Function):
    """RMSNorm that also hands x back for the block's residual ``x + attn(norm(x))`` (reference attn.py:226): x then has
    ONE consumer in the autograd graph, and the gradient arriving through the residual is added into dx by the RMSNorm
    backward kernel itself instead of a separate accumulation pass over [B*S, d]."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, eps: float, tap: bool = False):
        """``tap`` (extension): a THIRD alias of x for the U-ViT's long-range skip (reference attn.py:282-288: an encoder block's output
        feeds the next block and the mirrored decoder block) -- its gradient, too, is added by the backward kernel, so x keeps one
        consumer and the autograd engine runs no accumulation pass for it"""
        xc = x if x.is_contiguous() else x.contiguous()
        y, rstd, yb = ops.rmsnorm_fwd(xc, weight, eps, _want_bf16_copy(xc))
        ctx.save_for_backward(xc, weight, rstd)
        yb = yb if yb is not None else torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(yb)
        ctx.set_materialize_grads(False)   # an unused output (the bf16 copy, the residual) gets None, not a zero fill
        ctx.nargs = 4       # (apply is always called with the four arguments)
        if tap:
            return y, xc.detach(), yb, xc.detach()
        return y, xc.detach(), yb

    @staticmethod
    def backward(ctx, dy: Tensor, dres: Optional[Tensor], _dyb=None, dtap: Optional[Tensor] = None):
        x, w, rstd = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(x)
        d = dy if dy.is_contiguous() else dy.contiguous()
        dx, dw = ops.rmsnorm_bwd(x, w, d, rstd, None if dres is None else dres.reshape(x.shape), defer=ops.defer_ok((w,)),
                                 dx_add2=None if dtap is None else dtap.reshape(x.shape))
        return (dx, dw, None, None)[:ctx.nargs]


class AttentionFn(Function):
    """softmax(QK^T/sqrt(d))V on a fused [B*S, (h+2*hkv)*32] projection, optional 1-D RoPE on q,k, optional dropout
    on the attention weights (reference attn.py:122-127).
    precision fp32: exact-fp32 MFMA kernels (csrc/attn.hip); bf16: csrc/attn_bf16.hip."""

    @staticmethod
    def forward(ctx, qkv: Tensor, freqs: Optional[Tensor], b: int, s: int, h: int, hkv: int, dropout_p: float = 0.0,
                head_group=None, head0: int = 0, heads_total: int = 0):
        """``head_group`` (extension, gaot_3d_amd/sharding.py): a process group whose ranks hold the SAME qkv (replicated
        Transformer of a point-sharded sample); rank r computes heads [r*h/G, (r+1)*h/G) only and the outputs are
        all-gathered, so the attention work -- 60 % of a 500 K-point step -- is divided by G instead of repeated G times."""
        scale = 1.0 / (32 ** 0.5)
        bf16 = ops.get_precision() == "bf16"
        pre_img = getattr(qkv, "_gaot_qkv_image", None)   # MultiLinearFn wrote the projection as the kernels' image
        if pre_img is not None and not (bf16 and head_group is None):
            raise GaotError("a q|k|v image placeholder reached an attention path that needs the fp32 projection")
        if pre_img is None:
            qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        ctx.hp = None
        if head_group is not None:
            import torch.distributed as dist
            gsz, grk = dist.get_world_size(head_group), dist.get_rank(head_group)
            if gsz > 1 and h % gsz == 0 and hkv % gsz == 0:
                from .sharding import local_qkv
                ctx.hp = (head_group, gsz, grk, h, hkv)
                qkv = local_qkv(qkv, grk, gsz, h, hkv)
                h, hkv = h // gsz, hkv // gsz
        # ONE seed word for all ranks (same torch.manual_seed -> same device seed stream); the kernels key a head's mask by its
        # GLOBAL index (head0 + local head of heads_total), so a head draws the same mask on whichever rank it runs and a
        # sharded step reproduces the unsharded one with dropout on (head0 / heads_total: a sequence-parallel caller's slice)
        seed = next_dropout_seed(qkv.device) if dropout_p > 0.0 else None
        if ctx.hp is not None:
            head0, heads_total = ctx.hp[2] * h, ctx.hp[3]
        ctx.heads = (int(head0), int(heads_total))
        if bf16:
            o, lse, img = ops.attn_fwd_bf16(qkv, freqs, b, s, h, hkv, scale, dropout_p, seed, image=pre_img, head0=head0,
                                            heads_total=heads_total)
            keep = img
        else:
            if freqs is not None:
                qkv = qkv.clone()
                ops.rope_(qkv, b * s, qkv.shape[1], 0, h + hkv, s, freqs, False)  # q then k heads: adjacent columns
            o, lse = ops.attn_fwd(qkv, b, s, h, hkv, scale, dropout_p, seed, head0, heads_total)
            keep = qkv
        empty = torch.empty(0, device=qkv.device)
        ctx.save_for_backward(keep, o, lse, freqs if freqs is not None else empty, seed if seed is not None else empty)
        ctx.dims = (b, s, h, hkv, scale, freqs is not None, bf16, dropout_p)
        if ctx.hp is not None:
            from .sharding import gather_head_outputs
            return gather_head_outputs(o, ctx.hp[0], ctx.hp[1])                    # heads back in global order
        return o

    @staticmethod
    def backward(ctx, d_o: Tensor):
        keep, o, lse, freqs, seed = ctx.saved_tensors
        b, s, h, hkv, scale, rope, bf16, dropout_p = ctx.dims
        seed = seed if dropout_p > 0.0 else None
        if ctx.hp is not None:       # this rank's heads of the (replicated) output gradient
            grk = ctx.hp[2]
            d_o = d_o[:, grk * h * 32:(grk + 1) * h * 32]
        pre = getattr(d_o, "_gaot_do_image", None)     # BlockTailFn.backward: dO already is the kernels' bf16 image, delta is formed
        if pre is not None and not (bf16 and ctx.hp is None):
            raise GaotError("a dO image placeholder reached an attention backward that needs the fp32 gradient")
        if pre is not None:
            dqkv = ops.attn_bwd_bf16(keep, o, None, lse, b, s, h, hkv, scale, dropout_p, seed, freqs if rope else None,
                                     do_image=pre[0], delta=pre[1], head0=ctx.heads[0], heads_total=ctx.heads[1])
            return dqkv, None, None, None, None, None, None, None, None, None
        d = d_o if d_o.is_contiguous() else d_o.contiguous()
        if bf16:   # inverse RoPE of dq / dk happens in the kernels' epilogues (-0.15 ms per step against a separate pass)
            dqkv = ops.attn_bwd_bf16(keep, o, d, lse, b, s, h, hkv, scale, dropout_p, seed, freqs if rope else None,
                                     head0=ctx.heads[0], heads_total=ctx.heads[1])
        else:
            dqkv = ops.attn_bwd(keep, o, d, lse, b, s, h, hkv, scale, dropout_p, seed, ctx.heads[0], ctx.heads[1])
            if rope:
                ops.rope_(dqkv, b * s, dqkv.shape[1], 0, h + hkv, s, freqs, True)
        if ctx.hp is not None:
            from .sharding import gather_qkv_grads
            group, gsz, grk, hg, kg = ctx.hp
            dqkv = gather_qkv_grads(dqkv, group, gsz, hg, kg)
        return dqkv, None, None, None, None, None, None, None, None, None


class DropoutFn(Function):
    """nn.Dropout in training mode (reference mlp.py:268-272, 318-322): counter-based mask from one word of the device seed
    stream, regenerated for the gradient"""

    @staticmethod
    def forward(ctx, x: Tensor, p: float, salt: int = 0, base_seed: Optional[Tensor] = None):
        xc = x if x.is_contiguous() else x.contiguous()
        seed = base_seed if base_seed is not None else next_dropout_seed(xc.device)
        if salt:      # a sub-stream of the seed word: a rank of a sharded step, or a (global) head of one attention call
            seed = seed + salt * 0x632BE59BD9B4E019 % (1 << 63)
        ctx.save_for_backward(seed)
        ctx.p = p
        return ops.dropout(xc, seed, p)

    @staticmethod
    def backward(ctx, d: Tensor):
        (seed,) = ctx.saved_tensors
        return ops.dropout(d if d.is_contiguous() else d.contiguous(), seed, ctx.p), None, None, None


def dropout(x: Tensor, p: float, training: bool, salt: int = 0, base_seed: Optional[Tensor] = None) -> Tensor:
    return DropoutFn.apply(x, float(p), int(salt), base_seed) if (training and p > 0.0) else x


class MatmulFn(Function):
    """C = A B (trans_b False) or A B^T (True) on contiguous 2-D fp32 matrices, exact-fp32 MFMA GEMM, with its autograd"""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, trans_b: bool):
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        m, k = a.shape
        n = b.shape[0] if trans_b else b.shape[1]
        ctx.save_for_backward(a, b)
        ctx.trans_b = trans_b
        return ops.gemm(a, b, m, n, k, k, b.shape[1], False, trans_b, precision=0)

    @staticmethod
    def backward(ctx, dc: Tensor):
        a, b = ctx.saved_tensors
        dc = dc if dc.is_contiguous() else dc.contiguous()
        m, k = a.shape
        n = dc.shape[1]
        if ctx.trans_b:   # C = A B^T, B [n, k]: dA = dC B ; dB = dC^T A
            da = ops.gemm(dc, b, m, k, n, n, k, False, False, precision=0)
            db = ops.gemm(dc, a, n, k, m, n, k, True, False, precision=0)
        else:             # C = A B, B [k, n]: dA = dC B^T ; dB = A^T dC
            da = ops.gemm(dc, b, m, k, n, n, n, False, True, precision=0)
            db = ops.gemm(a, dc, k, n, m, k, n, True, False, precision=0)
        return da, db, None


class RowSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, s: Tensor):
        w = ops.row_softmax(s if s.is_contiguous() else s.contiguous())
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, dw: Tensor):
        (w,) = ctx.saved_tensors
        return ops.row_softmax(w, grad=dw if dw.is_contiguous() else dw.contiguous(), w=w)


class RopeFn(Function):
    """1-D RoPE over the flattened token index on the q and k heads of a fused projection, any even head_dim"""

    @staticmethod
    def forward(ctx, qkv: Tensor, freqs: Tensor, rows: int, s: int, nheads: int, head_dim: int):
        out = qkv.clone()
        ops.rope_(out, rows, out.shape[1], 0, nheads, s, freqs, False, head_dim)
        ctx.save_for_backward(freqs)
        ctx.meta = (rows, s, nheads, head_dim)
        return out

    @staticmethod
    def backward(ctx, d: Tensor):
        (freqs,) = ctx.saved_tensors
        rows, s, nheads, head_dim = ctx.meta
        g = d.clone()
        ops.rope_(g, rows, g.shape[1], 0, nheads, s, freqs, True, head_dim)
        return g, None, None, None, None, None


def attention_general(qkv: Tensor, freqs: Optional[Tensor], b: int, s: int, h: int, hkv: int, head_dim: int,
                      dropout_p: float = 0.0, head0: int = 0, heads_total: int = 0) -> Tensor:
    """softmax(Q K^T / sqrt(d)) V for ANY head_dim (reference attn.py:110-127 accepts every hidden_size % num_heads == 0):
    the unfused general path -- per (batch, head) an S x S score matrix in HBM, exact-fp32 MFMA GEMMs, the row softmax and
    element dropout kernels, autograd by composition.  head_dim 32 (every shipped configuration) runs the flash kernels."""
    from . import edgeops as EO
    rows = b * s
    if freqs is not None:
        qkv = RopeFn.apply(qkv, freqs, rows, s, h + hkv, head_dim)
    scale = torch.full((s,), 1.0 / (head_dim ** 0.5), dtype=torch.float32, device=qkv.device)
    rep = h // hkv
    outs = []
    # one word of the seed stream per attention call; (batch, GLOBAL head) selects its sub-stream, so a head of a head- /
    # sequence-parallel rank (heads head0 .. of heads_total) draws the mask it draws in the unsharded step
    base = next_dropout_seed(qkv.device) if dropout_p > 0.0 else None
    htot = heads_total if heads_total > 0 else h
    for bi in range(b):
        blk = qkv[bi * s:(bi + 1) * s]
        heads = []
        for hi in range(h):
            kv = hi // rep
            q = EO.RowScaleFn.apply(blk[:, hi * head_dim:(hi + 1) * head_dim].contiguous(), scale)
            k = blk[:, (h + kv) * head_dim:(h + kv + 1) * head_dim]
            v = blk[:, (h + hkv + kv) * head_dim:(h + hkv + kv + 1) * head_dim]
            p = RowSoftmaxFn.apply(MatmulFn.apply(q, k, True))
            p = dropout(p, dropout_p, dropout_p > 0.0, 1 + bi * htot + head0 + hi, base)
            heads.append(MatmulFn.apply(p, v, False))
        outs.append(torch.cat(heads, dim=1))
    return outs[0] if b == 1 else torch.cat(outs, dim=0)


class SwiGLUFn(Function):
    @staticmethod
    def forward(ctx, ag: Tensor, f: int):
        ctx.save_for_backward(ag)
        ctx.f = f
        return ops.swiglu_fwd(ag, f)

    @staticmethod
    def backward(ctx, du: Tensor):
        (ag,) = ctx.saved_tensors
        d = du if du.is_contiguous() else du.contiguous()
        return ops.swiglu_bwd(ag, d, ctx.f), None


_FFN_BWD_FUSED = os.environ.get("GAOT_FFN_BWD_FUSED", "0") == "1"


class FFNFn(Function):
    """w2(silu(w1 x) * w3 x) [+ residual] (reference FFN.forward, attn.py:150-157) for the bf16 path with the
    intermediates kept as bf16 in memory: [rows, 2F] = w1 x | w3 x, silu(a)*g and both of their gradients are written
    once as bf16 by their producer (GEMM epilogue / SwiGLU kernel) and read as bf16 by their consumers -- they are the
    widest tensors of a block (134 MB each as fp32 at S = 16 384, F = 1024) and every GEMM that touches them rounds them
    to bf16 anyway.  Needs w1 | w3 co-located (one [2F, d] matrix)."""

    @staticmethod
    def eligible(x: Tensor, w1: Tensor, w3: Tensor, w2: Tensor) -> bool:
        f, d = w1.shape
        return (ops.get_precision() == "bf16" and x.is_cuda and d > 64 and d % 8 == 0 and f % 8 == 0 and 2 * f > 64
                and w3.shape == w1.shape and tuple(w2.shape) == (d, f) and _adjacent([_w2d(w1), _w2d(w3)])
                and all(t.requires_grad for t in (w1, w3, w2)))

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, w3: Tensor, w2: Tensor, residual: Optional[Tensor], res_is_x: bool = False):
        """``res_is_x``: the residual IS the input (the block's ``h + ffn(h)``, attn.py:229): its gradient is folded into the
        dx GEMM's epilogue instead of meeting dx in a separate accumulation pass"""
        f, d = w1.shape
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        m = x2.shape[0]
        wcat32 = w1.new_empty(0).set_(w1.untyped_storage(), w1.storage_offset(), (2 * f, d), (d, 1))
        wcat = _wb(wcat32, 1)
        w2c = _wb(_w2d(w2), 1)
        wcat_t, w2t = _wbt(wcat32), _wbt(_w2d(w2))   # bf16 transposes from the per-forward cast pass, or None
        xb = bf16_copy_of(x, (m, d))
        xa = xb if xb is not None else x2          # bf16 image written by the producing RMSNorm
        res = None
        if res_is_x:
            res = x2
        elif residual is not None:
            res = residual.reshape(m, d)
            if not res.is_contiguous():
                res = res.contiguous()
        w2f = _w2d(w2)
        if xa.dtype == torch.bfloat16 and _ffn_fusable(wcat32, w2f) and (res is None or res.dtype == torch.float32):
            # the whole FFN in ONE launch over 64-row blocks (csrc/ffn_fused.hip): u never returns from HBM; a | g and u are still
            # written for the backward below; bit-identical to the two launches of the other branch
            # Nothing is saved for the backward but the (bf16) input: FFNFn.backward recomputes a | g inside gaot_ffn_bwd_dag -- the 96 MB
            # of a | g and u per layer at S = 16 384 cross HBM once (backward) instead of three times.  Bit-identical to the other branch.
            need_bwd = any(ctx.needs_input_grad[:4])
            packed, has_bwd = _FFN_PACK_CACHE.get((wcat32.data_ptr(), tuple(wcat32.shape)), (None, False))
            if packed is None or (need_bwd and not has_bwd):
                packed = ops.ffn_pack(wcat32, w2f, f, need_bwd)
            y, _ag, _u = ops.ffn_fwd(xa, packed, f, res, save=False)
            empty = w2c.new_empty(0)
            ctx.save_for_backward(xa, wcat, w2c, packed, empty, wcat_t if wcat_t is not None else empty, w2t if w2t is not None else empty)
            ctx.fused = True
            ctx.res_is_x = res_is_x
            ctx.wparams = (w1, w3, w2)
            ctx.meta = (f, d, x.shape, residual.shape if (residual is not None and not res_is_x) else None, w1.shape, w2.shape)
            return y.view(*x.shape[:-1], d)
        else:
            if xa.dtype == torch.bfloat16 and wcat.dtype == torch.bfloat16 and d == 256 and f % 32 == 0:
                ag, u = ops.ffn_w13_swiglu(xa, wcat, f)     # projection + SwiGLU in one launch (csrc/gemm_k256.hip, OUT_SWIGLU)
            else:
                ag = ops.gemm(xa, wcat, m, 2 * f, d, d, d, False, True, precision=1, out_dtype=torch.bfloat16)
                u = ops.swiglu_fwd_bf16(ag, f)
            y = ops.gemm(u, w2c, m, d, f, f, f, False, True, residual=res, ldr=d, precision=1)
        empty = w2c.new_empty(0)
        ctx.save_for_backward(xa, wcat, w2c, ag, u, wcat_t if wcat_t is not None else empty, w2t if w2t is not None else empty)
        ctx.fused = False
        ctx.res_is_x = res_is_x
        ctx.wparams = (w1, w3, w2)
        ctx.meta = (f, d, x.shape, residual.shape if (residual is not None and not res_is_x) else None, w1.shape, w2.shape)
        return y.view(*x.shape[:-1], d)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, wcat, w2c, ag, u, wcat_t, w2t = ctx.saved_tensors
        f, d, xshape, rshape, w1shape, w2shape = ctx.meta
        m = x2.shape[0]
        dy2 = dy.reshape(m, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.fused:
            # (x2, wcat, w2c, packed, -, wcat_t, w2t): a | g recomputed, du = dy W2, the SwiGLU derivative and the bf16 copy of dy in ONE
            # launch (csrc/ffn_fused.hip: k_ffn_bwd); the three products that follow read its outputs
            packed = ag
            if ctx.needs_input_grad[0] and _FFN_BWD_DX:
                # ... and the input gradient dag W13 (+ dy) as well: the dag chunk is consumed on chip, the stand-alone K = 2F product is gone
                dx, dag, u, dyb = ops.ffn_bwd(x2, dy2, packed, f, ctx.res_is_x)
                dw2 = _dw_gemm(dyb, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
                dwcat = _dw_gemm(dag, x2, 2 * f, d, m, 2 * f, d, 1, ctx.wparams[:2])
                dres = dy2.view(rshape) if (rshape is not None and ctx.needs_input_grad[4]) else None
                return dx.view(xshape), dwcat[:f].view(w1shape), dwcat[f:].view(w1shape), dw2, dres, None
            dag, u, dyb = ops.ffn_bwd_dag(x2, dy2, packed, f)
            dw2 = _dw_gemm(dyb, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
        elif d == 256 and w2c.dtype == torch.bfloat16 and f % 64 == 0:
            # dy W2 as x W^T with both operands bf16 in memory and K = 256 -> the weights-in-registers kernel
            # (csrc/gemm_k256.hip: 41 -> 18 us at configs[1]) for one rounding pass over dy and a 0.5 MB weight transpose;
            # the weight-gradient GEMM reads the same bf16 rows (half the A traffic)
            dyb = ops.cast_bf16(dy2)
            if _FFN_BWD_FUSED:
                # measurement only (GAOT_FFN_BWD_FUSED=1): the SwiGLU backward in that product's epilogue (gaot_ffn_w2_bwd_swiglu: du
                # never reaches HBM).  Built, bit-compatible, and SLOWER: 54.9 us against 22.8 + 24.8 us at [16 384, 1024] -- a lane
                # of the transposed product owns a row, so the epilogue's reads of a | g touch 32 rows per instruction where the
                # stand-alone pass streams at 6.7 TB/s (profiles/archive/r5_t_ffn_bwd_fusion_lab.txt); the step: 21.98 against 21.93 ms
                dag = ops.ffn_w2_bwd_swiglu(dyb, w2t if w2t.numel() else w2c.t().contiguous(), ag, f)
            else:
                du = ops.gemm(dyb, w2t if w2t.numel() else w2c.t().contiguous(), m, f, d, d, d, False, True, precision=1,
                              out_dtype=torch.bfloat16)
                dag = ops.swiglu_bwd_bf16(ag, du, f)
            dw2 = _dw_gemm(dyb, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
        else:
            du = ops.gemm(dy2, w2c, m, f, d, d, f, False, False, precision=1, out_dtype=torch.bfloat16)
            dw2 = _dw_gemm(dy2, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
            dag = ops.swiglu_bwd_bf16(ag, du, f)
        dx = None
        if ctx.needs_input_grad[0]:
            if d == 256 and wcat.dtype == torch.bfloat16 and (2 * f) % 64 == 0 and 2 * f >= 512:
                # dag W13 as x W^T on the transposed bf16 weight [d, 2f]: both operands k-contiguous -> the streamed-weight
                # kernel (csrc/gemm_k256.hip: k_gemm_tn_n256) instead of the generic tile kernel with transposed reads
                dx = ops.gemm(dag, wcat_t if wcat_t.numel() else wcat.t().contiguous(), m, d, 2 * f, 2 * f, 2 * f, False, True,
                              residual=dy2 if ctx.res_is_x else None, ldr=d, precision=1).view(xshape)
            else:
                dx = ops.gemm(dag, wcat, m, d, 2 * f, 2 * f, d, False, False, residual=dy2 if ctx.res_is_x else None, ldr=d,
                              precision=1).view(xshape)
        dwcat = _dw_gemm(dag, x2, 2 * f, d, m, 2 * f, d, 1, ctx.wparams[:2])
        dres = dy2.view(rshape) if (rshape is not None and ctx.needs_input_grad[4]) else None
        return dx, dwcat[:f].view(w1shape), dwcat[f:].view(w1shape), dw2, dres, None


class NormFFNFn(Function):
    """RMSNorm -> FFN -> + the normalised input: the second half of a Transformer block (reference attn.py:227-229:
    ``h = ffn_norm(h); h + ffn(h)``) as ONE forward launch (csrc/ffn_fused.hip, NORM) -- the stand-alone norm pass and its fp32 output are
    gone; saved for the backward: h, bf16(norm(h)), 1/rms.  Backward: gaot_ffn_bwd (dx w.r.t. the normalised rows, the residual's
    gradient included), the two weight-gradient products, gaot_rmsnorm_bwd.  Values: those of RMSNormFn + FFNFn."""

    @staticmethod
    def eligible(h: Tensor, norm_w: Tensor, w1: Tensor, w3: Tensor, w2: Tensor) -> bool:
        if not (_FFN_FUSED and _NORM_FFN and h.is_cuda and h.dtype == torch.float32 and h.shape[-1] == 256 and norm_w.numel() == 256
                and _adjacent([_w2d(w1), _w2d(w3)]) and all(t.requires_grad for t in (w1, w3, w2)) and torch.is_grad_enabled()):
            return False
        f = w1.shape[0]
        wcat32 = w1.new_empty(0).set_(w1.untyped_storage(), w1.storage_offset(), (2 * f, 256), (256, 1))
        return _ffn_fusable(wcat32, _w2d(w2))

    @staticmethod
    def forward(ctx, h: Tensor, norm_w: Tensor, eps: float, w1: Tensor, w3: Tensor, w2: Tensor):
        f, d = w1.shape
        h2 = h.reshape(-1, d)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        wcat32 = w1.new_empty(0).set_(w1.untyped_storage(), w1.storage_offset(), (2 * f, d), (d, 1))
        packed, has_bwd = _FFN_PACK_CACHE.get((wcat32.data_ptr(), tuple(wcat32.shape)), (None, False))
        if packed is None or not has_bwd:
            packed = ops.ffn_pack(wcat32, _w2d(w2), f, True)
        y, yb, rstd = ops.norm_ffn_fwd(h2, norm_w, eps, packed, f)
        ctx.save_for_backward(h2, norm_w, rstd, yb, packed)
        ctx.wparams = (w1, w3, w2)
        ctx.nparam = norm_w
        ctx.meta = (f, d, h.shape, w1.shape, w2.shape)
        return y.view(*h.shape[:-1], d)

    @staticmethod
    def backward(ctx, dy: Tensor):
        h2, norm_w, rstd, yb, packed = ctx.saved_tensors
        f, d, hshape, w1shape, w2shape = ctx.meta
        m = h2.shape[0]
        dy2 = dy.reshape(m, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if _NORM_BWD_FUSED:
            dh, dag, u, dyb, dnw = ops.ffn_bwd_norm(yb, dy2, packed, f, h2, norm_w, rstd, defer=ops.defer_ok((ctx.nparam,)))
        else:
            dn, dag, u, dyb = ops.ffn_bwd(yb, dy2, packed, f, True)       # dn: gradient w.r.t. the normalised rows (FFN input + residual)
            dh, dnw = ops.rmsnorm_bwd(h2, norm_w, dn, rstd, defer=ops.defer_ok((ctx.nparam,)))
        dw2 = _dw_gemm(dyb, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
        dwcat = _dw_gemm(dag, yb, 2 * f, d, m, 2 * f, d, 1, ctx.wparams[:2])
        return dh.view(hshape), dnw, None, dwcat[:f].view(w1shape), dwcat[f:].view(w1shape), dw2


class NormQKVFn(Function):
    """attn_norm -> q | k | v projections -> RoPE, written straight as the attention kernels' bf16 image: the HEAD of a Transformer block
    (reference attn.py:104-109, 118-120, 226) as ONE forward launch (csrc/ffn_fused.hip: k_norm_qkv).  Outputs: the storage-less q | k | v
    placeholder that carries the image (as MultiLinearFn's), the residual alias of x and, with ``tap``, the skip alias (as RMSNormResFn's).
    Backward: the input-gradient and weight-gradient products on d(q | k | v), then gaot_rmsnorm_bwd(2) with the residual's and the
    skip's gradients as addends.  Values: those of RMSNormResFn + MultiLinearFn(image_spec)."""

    @staticmethod
    def eligible(x: Tensor, norm_w: Tensor, weights, image_spec) -> bool:
        ws = [_w2d(w) for w in weights]
        return (_NORM_QKV and ops.get_precision() == "bf16" and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256
                and image_spec is not None and norm_w.numel() == 256 and torch.is_grad_enabled() and _adjacent(ws)
                and sum(w.shape[0] for w in ws) % 256 == 0 and all(w.requires_grad for w in weights))

    @staticmethod
    def forward(ctx, x: Tensor, norm_w: Tensor, eps: float, tap: bool, image_spec, *weights: Tensor):
        ws = [_w2d(w) for w in weights]
        ntot = sum(w.shape[0] for w in ws)
        xc = x if x.is_contiguous() else x.contiguous()
        x2 = xc.reshape(-1, 256)
        m = x2.shape[0]
        wcat32 = ws[0].new_empty(0).set_(ws[0].untyped_storage(), ws[0].storage_offset(), (ntot, 256), (256, 1))
        packed = _QKV_PACK_CACHE.get((wcat32.data_ptr(), tuple(wcat32.shape)))
        if packed is None:
            packed = ops.qkv_pack_multi([wcat32], True)[0]
        freqs, b, s, h, hkv, scale = image_spec
        img, yb, rstd = ops.norm_qkv_image(x2, norm_w, eps, packed, b, s, h, hkv, freqs, scale)
        out = torch.empty(1, dtype=torch.float32, device=x.device).expand(m, ntot)   # shape only: no [m, ntot] buffer
        out._gaot_qkv_image = img
        ctx.save_for_backward(x2, norm_w, rstd, yb, _wb(wcat32, 1), packed)
        ctx.wparams, ctx.nparam = tuple(weights), norm_w
        ctx.meta = (x.shape, [w.shape for w in weights], ntot)
        ctx.set_materialize_grads(False)
        ctx.ntap = bool(tap)
        return (out, xc.detach(), xc.detach()) if tap else (out, xc.detach())

    @staticmethod
    def backward(ctx, dqkv: Tensor, dres: Optional[Tensor] = None, dtap: Optional[Tensor] = None):
        x2, norm_w, rstd, yb, wcat, packed = ctx.saved_tensors
        xshape, wshapes, ntot = ctx.meta
        m = x2.shape[0]
        d = dqkv if dqkv.is_contiguous() else dqkv.contiguous()
        dwcat = _dw_gemm(d, yb, ntot, 256, m, ntot, 256, 1, ctx.wparams)
        dres2 = None if dres is None else dres.reshape(x2.shape)
        dtap2 = None if dtap is None else dtap.reshape(x2.shape)
        if _NORM_BWD_FUSED:
            # d(norm x) = dqkv Wqkv with attn_norm's backward (and the residual's / the skip's addends) in its epilogue: one launch
            dx, dnw = ops.qkv_bwd_norm(d, packed, x2, norm_w, rstd, dres2, dtap2, defer=ops.defer_ok((ctx.nparam,)))
        else:
            dn = ops.gemm(d, wcat, m, 256, ntot, ntot, 256, False, False, precision=1)
            dx, dnw = ops.rmsnorm_bwd(x2, norm_w, dn, rstd, dres2, defer=ops.defer_ok((ctx.nparam,)), dx_add2=dtap2)
        dws, col = [], 0
        for shp in wshapes:
            dws.append(dwcat[col:col + shp[0]].view(shp))
            col += shp[0]
        return (dx.view(xshape), dnw, None, None, None, *dws)


class CatNormQKVFn(Function):
    """NormQKVFn with the decoder block's skip projection in front (reference attn.py:222-225: ``x = skip_proj(cat([x, skip]))``): ONE
    forward launch for the two projection GEMMs, attn_norm, q | k | v and RoPE (csrc/ffn_fused.hip: k_norm_qkv<CAT>).  Outputs: the
    q | k | v placeholder carrying the image and the residual alias of the PROJECTED rows.  Backward: NormQKVFn's, then CatLinearFn's on
    the projected rows' gradient (the same tensor given as x and as skip folds its two gradients in one GEMM epilogue)."""

    @staticmethod
    def eligible(x: Tensor, skip: Tensor, wskip: Tensor, norm_w: Tensor, weights, image_spec) -> bool:
        return (_CAT_QKV and tuple(_w2d(wskip).shape) == (256, 512) and wskip.dtype == torch.float32 and wskip.requires_grad and skip.shape == x.shape
                and skip.dtype == torch.float32 and NormQKVFn.eligible(x, norm_w, weights, image_spec))

    @staticmethod
    def forward(ctx, x: Tensor, skip: Tensor, wskip: Tensor, bskip: Optional[Tensor], norm_w: Tensor, eps: float, image_spec, *weights: Tensor):
        ws = [_w2d(w) for w in weights]
        ntot = sum(w.shape[0] for w in ws)
        ctx.same = _same_view(x, skip)
        xa = x.reshape(-1, 256)
        xb = skip.reshape(-1, 256)
        xa = xa if xa.is_contiguous() else xa.contiguous()
        xb = xb if xb.is_contiguous() else xb.contiguous()
        m = xa.shape[0]
        wsk = _w2d(wskip)
        wcat32 = ws[0].new_empty(0).set_(ws[0].untyped_storage(), ws[0].storage_offset(), (ntot, 256), (256, 1))
        packed = _QKV_PACK_CACHE.get((wcat32.data_ptr(), tuple(wcat32.shape)))
        if packed is None:
            packed = ops.qkv_pack_multi([wcat32], True)[0]
        spk = _SKIP_PACK_CACHE.get((wsk.data_ptr(), tuple(wsk.shape)))
        if spk is None:
            spk = ops.skip_pack_multi([wsk])[0]
        freqs, b, s, h, hkv, scale = image_spec
        img, xo, yb, rstd = ops.cat_norm_qkv_image(xa, xb, spk, bskip, norm_w, eps, packed, b, s, h, hkv, freqs, scale)
        out = torch.empty(1, dtype=torch.float32, device=x.device).expand(m, ntot)   # shape only
        out._gaot_qkv_image = img
        ctx.save_for_backward(xa, xb, xo, norm_w, rstd, yb, _wb(wcat32, 1), packed, _wb(wsk, 1), spk)
        ctx.wparams, ctx.nparam, ctx.sparams = tuple(weights), norm_w, (wskip, bskip)
        ctx.meta = (x.shape, skip.shape, [w.shape for w in weights], ntot, wskip.shape)
        ctx.set_materialize_grads(False)
        return out, xo.view(x.shape)

    @staticmethod
    def backward(ctx, dqkv: Tensor, dres: Optional[Tensor] = None):
        xa, xb, xo, norm_w, rstd, yb, wcat, packed, wskb, spk = ctx.saved_tensors
        xshape, sshape, wshapes, ntot, wsshape = ctx.meta
        m = xa.shape[0]
        d = dqkv if dqkv.is_contiguous() else dqkv.contiguous()
        dwcat = _dw_gemm(d, yb, ntot, 256, m, ntot, 256, 1, ctx.wparams)
        dres2 = None if dres is None else dres.reshape(m, 256)
        dxa = dxb = None
        fused_dx = _NORM_BWD_FUSED and _CAT_BWD_DX and ctx.needs_input_grad[0] and (ctx.same or ctx.needs_input_grad[1])
        if fused_dx:    # the projection's two input gradients leave the same launch (k_qkv_bwd_norm<CATB>)
            dxo, dxa, dxb, dnw = ops.qkv_bwd_norm_cat(d, packed, xo, norm_w, rstd, dres2, spk, ctx.same, defer=ops.defer_ok((ctx.nparam,)))
        elif _NORM_BWD_FUSED:
            dxo, dnw = ops.qkv_bwd_norm(d, packed, xo, norm_w, rstd, dres2, None, defer=ops.defer_ok((ctx.nparam,)))
        else:
            dn = ops.gemm(d, wcat, m, 256, ntot, ntot, 256, False, False, precision=1)
            dxo, dnw = ops.rmsnorm_bwd(xo, norm_w, dn, rstd, dres2, defer=ops.defer_ok((ctx.nparam,)))
        # the skip projection's backward (CatLinearFn.backward on [xa | xb] W^T + b)
        n, k = 256, 512
        wsk, bsk = ctx.sparams
        if not fused_dx:
            dxa = ops.gemm(dxo, wskb, m, 256, n, n, k, False, False, precision=1) if (ctx.needs_input_grad[0] or ctx.same) else None
            if ctx.same and ctx.needs_input_grad[0]:
                dxa = ops.gemm(dxo, wskb[:, 256:], m, 256, n, n, k, False, False, residual=dxa, ldr=256, precision=1)
            else:
                dxb = ops.gemm(dxo, wskb[:, 256:], m, 256, n, n, k, False, False, precision=1) if ctx.needs_input_grad[1] else None
        dws = torch.empty(n, k, dtype=torch.float32, device=d.device)
        ops.gemm(dxo, xa, n, 256, m, n, 256, True, False, out=dws, ldc=k, precision=1)
        ops.gemm(dxo, xb, n, 256, m, n, 256, True, False, out=dws[:, 256:], ldc=k, precision=1)
        dbs = ops.colsum(dxo, m, n, n) if (bsk is not None and ctx.needs_input_grad[3]) else None
        dwl, col = [], 0
        for shp in wshapes:
            dwl.append(dwcat[col:col + shp[0]].view(shp))
            col += shp[0]
        return (None if dxa is None else dxa.view(xshape), None if dxb is None else dxb.view(sshape), dws.view(wsshape), dbs, dnw, None, None, *dwl)


class BlockTailFn(Function):
    """Everything of a Transformer block behind the attention kernels (reference attn.py:127, 226-229):
    ``h = x + o_proj(attn_out); n = ffn_norm(h); out = n + ffn(n)`` as ONE forward launch (csrc/ffn_fused.hip, OPROJ + NORM).
    Saved: attn_out, h, bf16(n), 1/rms.  Backward: gaot_ffn_bwd, the two FFN weight-gradient products, gaot_rmsnorm_bwd, then the o_proj
    input- and weight-gradient products on dh; the gradient of x is dh."""

    @staticmethod
    def enabled() -> bool:
        return _FFN_FUSED and _NORM_FFN and _BLOCK_TAIL and ops.get_precision() == "bf16"

    @staticmethod
    def eligible(x: Tensor, wo: Tensor, norm_w: Tensor, w1: Tensor, w3: Tensor, w2: Tensor) -> bool:
        wo2 = _w2d(wo)
        return (BlockTailFn.enabled() and tuple(wo2.shape) == (256, 256) and wo2.dtype == torch.float32 and wo.requires_grad
                and NormFFNFn.eligible(x, norm_w, w1, w3, w2))

    @staticmethod
    def forward(ctx, o: Tensor, x: Tensor, wo: Tensor, norm_w: Tensor, eps: float, w1: Tensor, w3: Tensor, w2: Tensor, attn_dims=None):
        """``attn_dims`` = (b, s, heads, kv heads) when ``o`` comes straight out of AttentionFn's bf16 kernels (8 heads of 32, unsharded):
        the backward then hands its gradient over as the flash backward's bf16 dO image + row constants (gaot_oproj_bwd_image) on a
        shape-only placeholder -- the fp32 d_o and the preparation pass over it never exist"""
        ctx.attn_dims = attn_dims if (attn_dims is not None and _OPROJ_BWD_IMAGE and attn_dims[2] == 8 and attn_dims[0] * attn_dims[1] == o.shape[0]) else None
        f, d = w1.shape
        o2 = o if o.is_contiguous() else o.contiguous()
        x2 = x if x.is_contiguous() else x.contiguous()
        wo2 = _w2d(wo)
        wcat32 = w1.new_empty(0).set_(w1.untyped_storage(), w1.storage_offset(), (2 * f, d), (d, 1))
        key = (wcat32.data_ptr(), tuple(wcat32.shape))
        packed, _has_bwd = _FFN_PACK_CACHE.get(key, (None, False))
        if packed is None or _WO_PACKED.get(key) != wo2.data_ptr():
            packed = ops.block_pack_multi([(wcat32, _w2d(w2), wo2)], f)[0]
        y, h, yb, rstd = ops.block_tail_fwd(o2, x2, norm_w, eps, packed, f)
        ctx.save_for_backward(o2, _wb(wo2, 1), h, norm_w, rstd, yb, packed)
        ctx.wparams = (w1, w3, w2)
        ctx.nparam, ctx.oparam = norm_w, wo
        ctx.meta = (f, d, w1.shape, w2.shape, wo.shape)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        o2, wob, h, norm_w, rstd, yb, packed = ctx.saved_tensors
        f, d, w1shape, w2shape, woshape = ctx.meta
        m = h.shape[0]
        dy2 = dy if dy.is_contiguous() else dy.contiguous()
        if _NORM_BWD_FUSED:
            dh, dag, u, dyb, dnw = ops.ffn_bwd_norm(yb, dy2, packed, f, h, norm_w, rstd, defer=ops.defer_ok((ctx.nparam,)))
        else:
            dn, dag, u, dyb = ops.ffn_bwd(yb, dy2, packed, f, True)
            dh, dnw = ops.rmsnorm_bwd(h, norm_w, dn, rstd, defer=ops.defer_ok((ctx.nparam,)))
        dw2 = _dw_gemm(dyb, u, d, f, m, d, f, 1, ctx.wparams[2:]).view(w2shape)
        dwcat = _dw_gemm(dag, yb, 2 * f, d, m, 2 * f, d, 1, ctx.wparams[:2])
        d_o = None
        if ctx.needs_input_grad[0]:
            if ctx.attn_dims is not None:
                b_, s_, h_, hkv_ = ctx.attn_dims
                img, delta = ops.oproj_bwd_image(dh, o2, packed, f, b_, s_, h_, hkv_)
                d_o = torch.empty(1, dtype=torch.float32, device=dh.device).expand(m, d)      # shape only
                d_o._gaot_do_image = (img, delta)
            else:
                d_o = ops.gemm(dh, wob, m, d, d, d, d, False, False, precision=1)
        dwo = _dw_gemm(dh, o2, d, d, m, d, d, 1, (ctx.oparam,)).view(woshape)
        return d_o, (dh if ctx.needs_input_grad[1] else None), dwo, dnw, None, dwcat[:f].view(w1shape), dwcat[f:].view(w1shape), dw2, None


class Mlp2Fn(Function):
    """Per-node two-layer MLP  W2 gelu(W1 x + b1) + b2  with the hidden layer kept in registers (csrc/mlp2.hip): the
    decoder's projection C -> 256 -> out (reference magno.py:793-797), bf16 path.  The unfused chain moves the
    [N, 256] fp32 hidden tensor through HBM eight times per step."""

    @staticmethod
    def eligible(x: Tensor, fcs, non_linearity: str) -> bool:
        if len(fcs) != 2 or non_linearity != "gelu" or ops.get_precision() != "bf16" or not x.is_cuda:
            return False
        w1, w2 = _w2d(fcs[0].weight), _w2d(fcs[1].weight)
        return (x.shape[-1] == 32 and w1.shape[1] == 32 and w1.shape[0] in (64, 128, 256) and 1 <= w2.shape[0] <= 4
                and w2.shape[1] == w1.shape[0] and fcs[0].bias is not None and x.dtype == torch.float32)

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Optional[Tensor]):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w1c, w2c = _w2d(w1), _w2d(w2)
        out = ops.mlp2_forward(x2, w1c, b1, w2c, b2)
        ctx.save_for_backward(x2, w1c, b1, w2c)
        ctx.meta = (x.shape, w1.shape, w2.shape, b2 is not None)
        return out.view(*x.shape[:-1], w2c.shape[0])

    @staticmethod
    def backward(ctx, dout: Tensor):
        x2, w1c, b1, w2c = ctx.saved_tensors
        xshape, w1shape, w2shape, has_b2 = ctx.meta
        d = dout.reshape(x2.shape[0], w2c.shape[0])
        if not d.is_contiguous():
            d = d.contiguous()
        dx, dw1, db1, dw2 = ops.mlp2_backward(x2, w1c, b1, w2c, d)
        db2 = ops.colsum(d, d.shape[0], d.shape[1], d.shape[1]) if has_b2 else None
        return dx.view(xshape), dw1.view(w1shape), db1, dw2.view(w2shape), db2


class AddFn(Function):
    """a + b (b optionally broadcast over leading rows with `period` elements)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, period: Optional[int]):
        ac = a if a.is_contiguous() else a.contiguous()
        bc = b if b.is_contiguous() else b.contiguous()
        ctx.period = period
        return ops.axpy(ac, bc, 1.0, period)

    @staticmethod
    def backward(ctx, g: Tensor):
        if ctx.period is not None and ctx.needs_input_grad[1]:
            raise GaotError("AddFn: gradient of a broadcast addend is not needed on the hot path")
        return g, (g if ctx.period is None else None), None


def add(a: Tensor, b: Tensor) -> Tensor:
    return AddFn.apply(a, b, None)


class PatchifyFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, b, d, h, w, p, c, to_tokens: bool):
        ctx.args = (b, d, h, w, p, c, to_tokens)
        xc = x if x.is_contiguous() else x.contiguous()
        return ops.patchify(xc, b, d, h, w, p, c, to_tokens)

    @staticmethod
    def backward(ctx, g: Tensor):
        b, d, h, w, p, c, to_tokens = ctx.args
        gc = g if g.is_contiguous() else g.contiguous()
        return ops.patchify(gc, b, d, h, w, p, c, not to_tokens), None, None, None, None, None, None, None


class MSELossFn(Function):
    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor):
        p = pred if pred.is_contiguous() else pred.contiguous()
        t = target if target.is_contiguous() else target.contiguous()
        if p.shape != t.shape:
            raise GaotError(f"mse_loss: shape mismatch {tuple(p.shape)} vs {tuple(t.shape)}")
        ctx.save_for_backward(p, t)
        return ops.mse_fwd(p, t)

    @staticmethod
    def backward(ctx, g: Tensor):
        p, t = ctx.saved_tensors
        gl = g if g.is_contiguous() else g.contiguous()
        return ops.mse_bwd(p, t, gl), None


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    """nn.MSELoss() stand-in (reference src/trainer/base.py:56, used at stat.py:550)."""
    return MSELossFn.apply(pred, target)


def _adjacent(ws) -> bool:
    """True when the 2-D weights sit back to back in one storage (same K): they then form ONE [sum N_i, K] matrix."""
    k = ws[0].shape[1]
    for a, b in zip(ws[:-1], ws[1:]):
        if b.shape[1] != k or not (a.is_contiguous() and b.is_contiguous()) or a.device != b.device:
            return False
        if b.data_ptr() != a.data_ptr() + a.numel() * a.element_size():
            return False
        if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
            return False   # neighbours in memory by allocator accident, not slices of one buffer
    return True


def colocate(params) -> None:
    """Re-point the storage of separate parameters (the reference keeps q_proj/k_proj/v_proj and w1/w3 as separate
    nn.Linear weights, attn.py:76-79,146-148) at consecutive slices of one buffer, values preserved, so that ONE GEMM
    serves all of them in forward, input-gradient and weight-gradient.  Parameter objects, names, shapes and the
    state_dict are unchanged; a later ``.to(device)`` simply undoes the co-location and the next forward redoes it."""
    ps = list(params)
    if _adjacent([_w2d(p.data) for p in ps]):
        return
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise GaotError("colocate: parameters must be co-located before the step is captured into a graph")
    with torch.no_grad():
        flat = torch.empty(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        off = 0
        for p in ps:
            v = flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            off += p.numel()


class MultiLinearFn(Function):
    """[x W_0^T | x W_1^T | ...] written into the column blocks of ONE buffer (bias-free), so the
    separate q/k/v (and w1/w3) parameters of the reference feed one fused downstream kernel."""

    @staticmethod
    def forward(ctx, x: Tensor, precision: Optional[int], image_spec, *weights: Tensor):
        """``image_spec`` = (freqs or None, b, s, h, hkv, scale) (extension): the caller is the attention layer and will hand
        the result to AttentionFn only.  When the bf16 fast path applies, the projection is written straight as the attention
        kernels' bf16 image (csrc/gemm_k256.hip, OUT_QKV_IMAGE) and the returned tensor is a storage-less placeholder of the
        right shape that carries the image (``_gaot_qkv_image``): the fp32 q|k|v never exists in HBM."""
        ws = [_w2d(w) for w in weights]
        k = ws[0].shape[1]
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        m = x2.shape[0]
        ntot = sum(w.shape[0] for w in ws)
        ctx.fused = len(ws) > 1 and _adjacent(ws) and all(ctx.needs_input_grad[3 + i] for i in range(len(ws)))
        ctx.wparams = tuple(weights)
        if ctx.fused:   # the weights are slices of one buffer (colocate): one [ntot, k] matrix, one GEMM
            wcat = _wb(ws[0].new_empty(0).set_(ws[0].untyped_storage(), ws[0].storage_offset(), (ntot, k), (k, 1)), precision)
            xb = bf16_copy_of(x, (m, k)) if wcat.dtype == torch.bfloat16 else None
            xa = xb if xb is not None else x2      # bf16 image written by the producing RMSNorm: half the A traffic
            if image_spec is not None and xb is not None and k == 256 and ntot % 64 == 0:
                freqs, b, s, h, hkv, scale = image_spec
                img = ops.qkv_image(xb, wcat, m, b, s, h, hkv, freqs, scale)
                out = torch.empty(1, dtype=torch.float32, device=x.device).expand(m, ntot)   # shape only: no [m, ntot] buffer
                out._gaot_qkv_image = img
                ctx.save_for_backward(xa, wcat)
                ctx.precision, ctx.xshape, ctx.wshapes = precision, x.shape, [w.shape for w in weights]
                return out
            out = torch.empty(m, ntot, dtype=torch.float32, device=x.device)
            ops.gemm(xa, wcat, m, ntot, k, k, k, False, True, out=out, ldc=ntot, precision=precision)
            ctx.save_for_backward(xa, wcat)
        else:
            out = torch.empty(m, ntot, dtype=torch.float32, device=x.device)
            col = 0
            for w in ws:
                n = w.shape[0]
                ops.gemm(x2, w, m, n, k, k, k, False, True, out=out[:, col:], ldc=ntot, precision=precision)
                col += n
            ctx.save_for_backward(x2, *ws)
        ctx.precision, ctx.xshape, ctx.wshapes = precision, x.shape, [w.shape for w in weights]
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        x2, *ws = ctx.saved_tensors
        m, k = x2.shape
        d = dout if dout.is_contiguous() else dout.contiguous()
        ntot = d.shape[1]
        dx = None
        dws = []
        col = 0
        if ctx.fused:
            (wcat,) = ws
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(d, wcat, m, k, ntot, ntot, k, False, False, precision=ctx.precision)
            dwcat = _dw_gemm(d, x2, ntot, k, m, ntot, k, ctx.precision, ctx.wparams)
            for shp in ctx.wshapes:
                n = shp[0]
                dws.append(dwcat[col:col + n].view(shp))
                col += n
            return (dx.view(ctx.xshape) if dx is not None else None, None, None, *dws)
        for i, w in enumerate(ws):
            n = w.shape[0]
            blk = d[:, col:]
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(blk, w, m, k, n, ntot, k, False, False, residual=dx, ldr=k, precision=ctx.precision)
            dws.append(_dw_gemm(blk, x2, n, k, m, ntot, k, ctx.precision, ctx.wparams[i:i + 1]).view(ctx.wshapes[i])
                       if ctx.needs_input_grad[3 + i] else None)
            col += n
        return (dx.view(ctx.xshape) if dx is not None else None, None, None, *dws)


def multi_linear(x: Tensor, weights, precision: Optional[int] = None, image_spec=None) -> Tensor:
    return MultiLinearFn.apply(x, precision, image_spec, *weights)


def _same_view(a: Tensor, b: Tensor) -> bool:
    """the same autograd tensor, or two identical views of one base (x.reshape(..) taken twice): gradients handed to either reach
    the same elements of the same tensor"""
    if a is b:
        return True
    ba, bb = (a._base if a._base is not None else a), (b._base if b._base is not None else b)
    return ba is bb and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()


class CatLinearFn(Function):
    """y = [x_0 | x_1 | ...] W^T + b without materialising the concatenation (reference: torch.cat then
    nn.Linear -- magno.py:494,571-575,771-775): one GEMM per input against the matching column block of
    W, chained through the residual input; the weight gradient is written block by block."""

    @staticmethod
    def forward(ctx, weight: Tensor, bias: Optional[Tensor], precision: Optional[int], *xs: Tensor):
        w = _w2d(weight)
        n, k = w.shape
        xs2 = [x if x.is_contiguous() else x.contiguous() for x in xs]
        if sum(x.shape[1] for x in xs2) != k:
            raise GaotError(f"cat_linear: inputs have {sum(x.shape[1] for x in xs2)} features, weight expects {k}")
        m = xs2[0].shape[0]
        if all(x.shape[1] > 64 and x.shape[1] % 8 == 0 for x in xs2):
            w = _wb(w, precision)
        y = None
        col = 0
        for i, x in enumerate(xs2):
            ki = x.shape[1]
            y = ops.gemm(x, w[:, col:], m, n, ki, ki, k, False, True, bias if i == 0 else None, 0, residual=y, ldr=n,
                         precision=precision)
            col += ki
        ctx.save_for_backward(w, *xs2)
        ctx.meta = (precision, weight.shape, bias is not None)
        # were two of the inputs the same autograd tensor?  (then a None for the later one is "no further contribution")
        ctx.dups = any(_same_view(xs[i], xs[j]) for i in range(len(xs)) for j in range(i))
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        w, *xs = ctx.saved_tensors
        precision, wshape, has_bias = ctx.meta
        n, k = w.shape
        d = dy if dy.is_contiguous() else dy.contiguous()
        m = d.shape[0]
        dw = torch.empty(n, k, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[0] else None
        dxs = []
        col = 0
        first = {}      # storage of an input -> index of its first occurrence
        for i, x in enumerate(xs):
            ki = x.shape[1]
            key = (x.data_ptr(), tuple(x.shape))
            j = first.get(key)
            if ctx.needs_input_grad[3 + i] and j is not None and dxs[j] is not None and ctx.dups:
                # the SAME tensor twice (the first decoder block of the U-ViT gets the last encoder output as x AND as its skip,
                # attn.py:282-288): its two gradients meet in this GEMM's epilogue and leave as ONE -- no accumulation pass by the engine
                dxs[j] = ops.gemm(d, w[:, col:], m, ki, n, n, k, False, False, residual=dxs[j], ldr=ki, precision=precision)
                dxs.append(None)
            else:
                dxs.append(ops.gemm(d, w[:, col:], m, ki, n, n, k, False, False, precision=precision)
                           if ctx.needs_input_grad[3 + i] else None)
                first.setdefault(key, i)
            if dw is not None:
                ops.gemm(d, x, n, ki, m, n, ki, True, False, out=dw[:, col:], ldc=k, precision=precision)
            col += ki
        db = ops.colsum(d, m, n, n) if (has_bias and ctx.needs_input_grad[1]) else None
        return (dw.view(wshape) if dw is not None else None, db, None, *dxs)


def cat_linear(xs, weight, bias=None, precision: Optional[int] = None) -> Tensor:
    return CatLinearFn.apply(weight, bias, precision, *xs)


class ScaleMixFn(Function):
    """sum_s softmax(logits)_s * x_s   (reference magno.py:590-594)"""

    @staticmethod
    def forward(ctx, logits: Tensor, *xs: Tensor):
        xc = [x if x.is_contiguous() else x.contiguous() for x in xs]
        lg = logits if logits.is_contiguous() else logits.contiguous()
        out, w = ops.scale_mix_fwd(xc, lg)
        ctx.save_for_backward(w, *xc)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        w, *xs = ctx.saved_tensors
        d = dout if dout.is_contiguous() else dout.contiguous()
        dxs, dlog = ops.scale_mix_bwd(xs, w, d)
        return (dlog, *dxs)
