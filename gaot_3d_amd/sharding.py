"""Point-sharding of ONE sample across the GPUs of a node (new w.r.t. the reference, whose only
parallelism is sample-level DDP, src/trainer/stat.py:431-436).

Partition: physical points are split into `world` contiguous ranges; every edge lives with its PHYSICAL
endpoint (encoder: source, decoder: query), so edges, features, targets and predictions are disjoint across
ranks.  Latent coordinates, the latent Transformer and all parameters are replicated.

Exchange steps (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests):
  forward : encoder GNO   per-token SUM [M,C] and COUNT [M], one fused [M,C+1] all_reduce(SUM) -> mean = sum / max(count,1)
  backward: decoder       partial d(loss)/d(latent) [M,C]    -> all_reduce(SUM) before the (replicated, hence
            identical) Transformer backward
  after backward: gradients of the per-point / per-edge parameters (encoder.lifting, encoder.gno, decoder.*)
            are partial sums -> one flat all_reduce (~125 KB).  Per-token parameters (encoder.geoembed,
            encoder.recovery, patch_linear, processor.*) see identical full gradients on every rank.
Sequence-parallel Transformer (``parallel="seq"``, the default of ShardedStep): every row-wise operator of the latent
Transformer (patch_linear, RMSNorm, q|k|v / o_proj / SwiGLU / skip_proj GEMMs, residuals) runs on the rank's S/G token
rows only; around the attention kernels an all-to-all turns "my rows, all heads" into "all rows, my heads" and back
(8 heads <-> 8 GPUs; 4 all-to-alls of <= 6.3 MB per rank per layer at configs[1] instead of replicating 27 ms of work);
the processed rows are all-gathered into the replicated latent grid for the decoder (backward: reduce-scatter of the
decoder's partial latent gradient), and the Transformer's weight gradients -- now partial sums over the rank's rows --
join the flat gradient all-reduce (45 MB).
Head-parallel attention (``parallel="head"``): rank r runs the attention kernels for heads [r*H/G, (r+1)*H/G)
of the replicated Transformer -- forward all_gather of the head outputs [S, d/G] (16.8 MB total at configs[1]),
backward all_gather of d(q|k|v) [S, 3d/G] (50 MB total) per layer; everything else of the Transformer stays
replicated, so no weight-gradient exchange is needed.  With attention dropout every head's mask is drawn by the one
rank that owns it (same seed stream on all ranks; the masks differ from the unsharded run's, which keys them by the
global head index).
The geometry-only GeoEmbed statistics of the encoder are sums over each token's edges, which are spread over the
ranks: every rank reduces its own edges to additive fp64 moments [M,12] (count, sum d, sum d^2, sum u, sum u u^T),
one SUM all_reduce (12.6 MB) combines them and every rank finishes the features (csrc/geoembed.hip) -- no rank holds
the full geometry, and the per-step work stays proportional to the local points.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import comm
from .data import MeshBatch

Tensor = torch.Tensor


def shard_range(n: int, rank: int, world: int):
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def shard_batch(batch: MeshBatch, rank: int, world: int, num_latent: int) -> MeshBatch:
    """Rank-local view of a single-graph batch: points [lo,hi), their features/targets, and the edges whose
    physical endpoint is in the range, re-indexed to local point ids."""
    if getattr(batch, "num_graphs", 1) != 1:
        raise ValueError("point-sharding splits ONE sample; batch several samples with DDP instead")
    n = batch.pos.shape[0]
    lo, hi = shard_range(n, rank, world)
    out = MeshBatch()
    for k, v in batch.__dict__.items():
        if k.startswith("_"):
            continue
        if not torch.is_tensor(v):
            setattr(out, k, v)
            continue
        if k.startswith("encoder_edge_index"):
            m = (v[0] >= lo) & (v[0] < hi)
            e = v[:, m].clone()
            e[0] -= lo
            setattr(out, k, e)
        elif k.startswith("decoder_edge_index"):
            m = (v[1] >= lo) & (v[1] < hi)
            e = v[:, m].clone()
            e[1] -= lo
            setattr(out, k, e)
        elif k == "ptr":
            continue
        elif v.dim() >= 1 and v.shape[0] == n:
            setattr(out, k, v[lo:hi].contiguous())
        else:
            setattr(out, k, v)
    out.num_graphs = 1
    out.shard = (rank, world, lo, hi, n)
    out.ptr = torch.tensor([0, hi - lo], dtype=torch.long, device=batch.pos.device)
    return out


class GlobalSegmentMeanFn(torch.autograd.Function):
    """local per-row mean + local degree  ->  mean over the edges of ALL ranks.
    Backward receives the full gradient (the consumer is replicated) and returns its local share."""

    @staticmethod
    def forward(ctx, local_mean: Tensor, local_deg: Tensor, group):
        # ONE all-reduce of [M, C + 1]: the per-token sums (the kernel's mean times its own edge count) and the counts
        buf = torch.cat([local_mean * local_deg[:, None], local_deg[:, None]], dim=1)
        comm.run(lambda: dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group), (buf,), "all_reduce")
        dg = buf[:, -1].clamp(min=1.0)
        ctx.save_for_backward(local_deg / dg)
        return buf[:, :-1] / dg[:, None]

    @staticmethod
    def backward(ctx, g: Tensor):
        (ratio,) = ctx.saved_tensors
        return g * ratio[:, None], None, None


class GlobalSegmentMaxFn(torch.autograd.Function):
    """local per-row maximum (0 where the rank has no edge of the row) + local degree -> (maximum over the edges of ALL ranks
    with rows that have no edge anywhere = 0, global degree).  Backward: the gradient of a row / channel goes to the rank(s)
    whose local maximum IS the global one (shared equally if several ranks tie, as torch's amax does among tied elements);
    the local segment-max backward then routes it to the arg-max edge.  PointNet GeoEmbed of a point-sharded encoder
    (reference geoembed.py:211-216, pooling 'max')."""

    @staticmethod
    def forward(ctx, local_max: Tensor, local_deg: Tensor, group):
        has_local = (local_deg > 0)[:, None]
        glob = torch.where(has_local, local_max, torch.full_like(local_max, float("-inf")))
        mine = glob.clone()
        comm.run(lambda: dist.all_reduce(glob, op=dist.ReduceOp.MAX, group=group), (glob,), "all_reduce")
        deg = local_deg.clone()
        comm.run(lambda: dist.all_reduce(deg, op=dist.ReduceOp.SUM, group=group), (deg,), "all_reduce")
        owner = ((mine == glob) & has_local).to(local_max.dtype)
        nown = owner.clone()
        comm.run(lambda: dist.all_reduce(nown, op=dist.ReduceOp.SUM, group=group), (nown,), "all_reduce")
        ctx.save_for_backward(owner / nown.clamp(min=1.0))
        ctx.mark_non_differentiable(deg)
        return torch.where((deg > 0)[:, None], glob, torch.zeros_like(glob)), deg

    @staticmethod
    def backward(ctx, g: Tensor, _gdeg=None):
        (share,) = ctx.saved_tensors
        return g * share, None, None


class AllReduceGradFn(torch.autograd.Function):
    """identity in forward; SUM all-reduce of the gradient in backward (partial -> full)."""

    @staticmethod
    def forward(ctx, x: Tensor, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous().clone()
        group = ctx.group
        comm.run(lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group), (g,), "all_reduce")
        return g, None


# ---- head-parallel attention: the index arithmetic of the exchange (device independent; kernels live in functional.py)
def head_slices(rank: int, world: int, h: int, hkv: int, hd: int = 32):
    """column ranges of rank's q / k / v heads inside a fused [rows, (h + 2 hkv) * hd] projection"""
    hl, kl = h // world, hkv // world
    return ((rank * hl * hd, (rank + 1) * hl * hd), ((h + rank * kl) * hd, (h + (rank + 1) * kl) * hd),
            ((h + hkv + rank * kl) * hd, (h + hkv + (rank + 1) * kl) * hd))


def local_qkv(qkv: Tensor, rank: int, world: int, h: int, hkv: int, hd: int = 32) -> Tensor:
    (q0, q1), (k0, k1), (v0, v1) = head_slices(rank, world, h, hkv, hd)
    return torch.cat([qkv[:, q0:q1], qkv[:, k0:k1], qkv[:, v0:v1]], dim=1)


def all_gather_stack(t: Tensor, group, world: int) -> Tensor:
    """[world, *t.shape]: one all-gather (RCCL: all_gather_into_tensor; gloo, used by the tests: list form)"""
    t = t if t.is_contiguous() else t.contiguous()
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "nccl":
        comm.run(lambda: dist.all_gather_into_tensor(out, t, group=group), (out, t), "all_gather")
    else:
        parts = list(out.unbind(0))
        comm.run(lambda: dist.all_gather(parts, t, group=group), (out, t), "all_gather")
    return out


def gather_head_outputs(o_local: Tensor, group, world: int) -> Tensor:
    """[rows, (h/world)*32] per rank -> [rows, h*32] with the heads back in global order"""
    allo = all_gather_stack(o_local, group, world)
    return allo.permute(1, 0, 2).reshape(o_local.shape[0], world * o_local.shape[1])


def gather_qkv_grads(dqkv_local: Tensor, group, world: int, h: int, hkv: int, hd: int = 32) -> Tensor:
    """[rows, (hl + 2 kl)*hd] per rank (q | k | v of the rank's heads) -> [rows, (h + 2 hkv)*hd] in the fused layout"""
    hl, kl = h // world, hkv // world
    rows = dqkv_local.shape[0]
    allg = all_gather_stack(dqkv_local, group, world).permute(1, 0, 2)           # [rows, world, (hl + 2 kl) * hd]
    return torch.cat([allg[:, :, :hl * hd].reshape(rows, h * hd),
                      allg[:, :, hl * hd:(hl + kl) * hd].reshape(rows, hkv * hd),
                      allg[:, :, (hl + kl) * hd:].reshape(rows, hkv * hd)], dim=1)


class LocalHeadsFn(torch.autograd.Function):
    """replicated q|k|v [rows, (h + 2 hkv) * hd] -> this rank's heads [rows, (h/G + 2 hkv/G) * hd]; backward all-gathers the
    head gradients back into the fused layout (head-parallel attention on the general path, any head_dim)"""

    @staticmethod
    def forward(ctx, qkv: Tensor, group, h: int, hkv: int, hd: int):
        ctx.meta = (group, dist.get_world_size(group), h, hkv, hd)
        return local_qkv(qkv, dist.get_rank(group), ctx.meta[1], h, hkv, hd).contiguous()

    @staticmethod
    def backward(ctx, g: Tensor):
        group, world, h, hkv, hd = ctx.meta
        return gather_qkv_grads(g if g.is_contiguous() else g.contiguous(), group, world, h, hkv, hd), None, None, None, None


class GatherHeadsFn(torch.autograd.Function):
    """head outputs of this rank [rows, (h/G) * hd] -> all heads in global order [rows, h * hd]; backward keeps the rank's slice"""

    @staticmethod
    def forward(ctx, o: Tensor, group):
        ctx.meta = (dist.get_world_size(group), dist.get_rank(group), o.shape[1])
        return gather_head_outputs(o if o.is_contiguous() else o.contiguous(), group, ctx.meta[0])

    @staticmethod
    def backward(ctx, g: Tensor):
        world, rank, w = ctx.meta
        return g[:, rank * w:(rank + 1) * w].contiguous(), None


# ---- sequence-parallel Transformer: the exchange steps as autograd Functions (device independent) ----------------------
def _all_to_all(send: Tensor, group) -> Tensor:
    """block j of ``send`` [G, ...] goes to rank j; block i of the result came from rank i.  RCCL: all_to_all_single; the
    gloo group of the one-GPU tests cannot exchange device tensors this way, so they are staged through the host there"""
    send = send if send.is_contiguous() else send.contiguous()
    recv = torch.empty_like(send)
    _all_to_all_into(recv, send, group)
    return recv


def _all_to_all_into(recv: Tensor, send: Tensor, group) -> None:
    """the same into an existing buffer (e.g. the head of an attention image)"""
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        def staged():
            h = send.cpu()
            r = torch.empty_like(h)
            dist.all_to_all_single(r, h, group=group)
            recv.copy_(r)
        comm.run(staged, (recv, send), "all_to_all")
    else:
        comm.run(lambda: dist.all_to_all_single(recv, send, group=group), (recv, send), "all_to_all")


def _reduce_scatter_rows(t: Tensor, group, world: int, rank: int) -> Tensor:
    """sum over the ranks of ``t`` [world * r, ...], rank keeps its r rows (RCCL: reduce_scatter_tensor)"""
    t = t if t.is_contiguous() else t.contiguous()
    r = t.shape[0] // world
    if dist.get_backend(group) == "nccl":
        out = torch.empty((r,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        comm.run(lambda: dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=group), (out, t), "reduce_scatter")
        return out
    full = t.clone()
    comm.run(lambda: dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group), (full,), "all_reduce")
    return full[rank * r:(rank + 1) * r].clone()


def _all_gather_rows(t: Tensor, group, world: int) -> Tensor:
    return all_gather_stack(t, group, world).reshape((world * t.shape[0],) + tuple(t.shape[1:]))


class SliceRowsFn(torch.autograd.Function):
    """replicated [S, ...] -> this rank's rows [S/G, ...]; backward all-gathers the row gradients, so every rank continues
    the (replicated) backward upstream with the full gradient"""

    @staticmethod
    def forward(ctx, x: Tensor, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if x.shape[0] % world:
            raise ValueError(f"sequence-parallel: {x.shape[0]} token rows do not divide over {world} ranks")
        ctx.group, ctx.world = group, world
        r = x.shape[0] // world
        return x[rank * r:(rank + 1) * r].contiguous()

    @staticmethod
    def backward(ctx, g: Tensor):
        return _all_gather_rows(g, ctx.group, ctx.world), None


class AllGatherRowsFn(torch.autograd.Function):
    """this rank's rows [S/G, ...] -> replicated [S, ...]; backward: the consumers' gradients are partial sums (each rank's
    decoder sees its own points) -> reduce-scatter, every rank gets the full gradient of its rows"""

    @staticmethod
    def forward(ctx, x: Tensor, group):
        ctx.group, ctx.world, ctx.rank = group, dist.get_world_size(group), dist.get_rank(group)
        return _all_gather_rows(x, group, ctx.world)

    @staticmethod
    def backward(ctx, g: Tensor):
        return _reduce_scatter_rows(g, ctx.group, ctx.world, ctx.rank), None


def qkv_segments(h: int, hkv: int, world: int, hd: int = 32):
    """(first column, per-rank width) of the q, k and v head groups inside a fused [rows, (h + 2 hkv) * hd] projection"""
    hl, kl = h // world * hd, hkv // world * hd
    return [(0, hl), (h * hd, kl), ((h + hkv) * hd, kl)]


def _pack_heads(qkv: Tensor, world: int, h: int, hkv: int, hd: int = 32) -> Tensor:
    """fused [rows, (h + 2 hkv) * hd] -> [world, rows, (hl + 2 kl) * hd]: block j = q | k | v of rank j's heads"""
    rows = qkv.shape[0]
    hl, kl = h // world * hd, hkv // world * hd
    out = torch.empty(world, rows, hl + 2 * kl, dtype=qkv.dtype, device=qkv.device)
    if qkv.is_cuda:    # one HIP launch (csrc/rowops.hip: k_pack_heads); the index arithmetic below is the host restatement
        from . import ops
        ops.pack_heads(qkv if qkv.is_contiguous() else qkv.contiguous(), out, world, qkv_segments(h, hkv, world, hd), True)
        return out
    out[:, :, :hl] = qkv[:, :h * hd].view(rows, world, hl).permute(1, 0, 2)
    out[:, :, hl:hl + kl] = qkv[:, h * hd:(h + hkv) * hd].view(rows, world, kl).permute(1, 0, 2)
    out[:, :, hl + kl:] = qkv[:, (h + hkv) * hd:].view(rows, world, kl).permute(1, 0, 2)
    return out


def _unpack_heads(blocks: Tensor, world: int, h: int, hkv: int, hd: int = 32) -> Tensor:
    """inverse of _pack_heads"""
    rows = blocks.shape[1]
    hl, kl = h // world * hd, hkv // world * hd
    out = torch.empty(rows, (h + 2 * hkv) * hd, dtype=blocks.dtype, device=blocks.device)
    if blocks.is_cuda:
        from . import ops
        ops.pack_heads(out, blocks if blocks.is_contiguous() else blocks.contiguous(), world, qkv_segments(h, hkv, world, hd), False)
        return out
    out[:, :h * hd].view(rows, world, hl).copy_(blocks[:, :, :hl].permute(1, 0, 2))
    out[:, h * hd:(h + hkv) * hd].view(rows, world, kl).copy_(blocks[:, :, hl:hl + kl].permute(1, 0, 2))
    out[:, (h + hkv) * hd:].view(rows, world, kl).copy_(blocks[:, :, hl + kl:].permute(1, 0, 2))
    return out


class SeqToHeadsFn(torch.autograd.Function):
    """q|k|v of MY token rows, all heads [S/G, (h + 2 hkv) * 32]  ->  q|k|v of ALL token rows, my heads
    [S, (h/G + 2 hkv/G) * 32] (one all-to-all); backward is the inverse exchange of the gradient"""

    @staticmethod
    def forward(ctx, qkv: Tensor, group, h: int, hkv: int, hd: int = 32):
        world = dist.get_world_size(group)
        if h % world or hkv % world:
            raise ValueError(f"sequence-parallel attention: {h} / {hkv} heads do not divide over {world} ranks")
        ctx.meta = (group, world, h, hkv, hd)
        recv = _all_to_all(_pack_heads(qkv, world, h, hkv, hd), group)      # block i = rank i's rows
        return recv.view(world * qkv.shape[0], recv.shape[2])

    @staticmethod
    def backward(ctx, g: Tensor):
        group, world, h, hkv, hd = ctx.meta
        g = g if g.is_contiguous() else g.contiguous()
        recv = _all_to_all(g.view(world, g.shape[0] // world, g.shape[1]), group)   # block j = my rows, rank j's heads
        return _unpack_heads(recv, world, h, hkv, hd), None, None, None, None


class HeadsToSeqFn(torch.autograd.Function):
    """attention output of ALL rows, my heads [S, (h/G) * 32] -> MY rows, all heads [S/G, h * 32] (heads in global order)"""

    @staticmethod
    def forward(ctx, o: Tensor, group):
        world = dist.get_world_size(group)
        ctx.meta = (group, world)
        o = o if o.is_contiguous() else o.contiguous()
        rows = o.shape[0] // world
        recv = _all_to_all(o.view(world, rows, o.shape[1]), group)          # block j = my rows, rank j's heads
        if recv.is_cuda:
            from . import ops
            out = torch.empty(rows, world * o.shape[1], dtype=o.dtype, device=o.device)
            ops.pack_heads(out, recv, world, [(0, o.shape[1])], False)
            return out
        return recv.permute(1, 0, 2).reshape(rows, world * o.shape[1])

    @staticmethod
    def backward(ctx, g: Tensor):
        group, world = ctx.meta
        rows, hl = g.shape[0], g.shape[1] // world
        if g.is_cuda:
            from . import ops
            send = torch.empty(world, rows, hl, dtype=g.dtype, device=g.device)
            ops.pack_heads(g if g.is_contiguous() else g.contiguous(), send, world, [(0, hl)], True)
        else:
            send = g.view(rows, world, hl).permute(1, 0, 2).contiguous()
        recv = _all_to_all(send, group)                                     # block i = rank i's rows, my heads
        return recv.view(world * rows, hl), None


# parameters whose gradients are partial sums over a rank's points / edges (the per-edge PointNet MLP of an encoder-side
# GeoEmbed included; its fc layer, like the statistical GeoEmbed's MLP, sees replicated per-token inputs)
PARTIAL_GRAD_PREFIXES = ("encoder.lifting.", "encoder.gno.", "encoder.geoembed.pointnet_mlp.", "decoder.")
SEQ_PARTIAL_GRAD_PREFIXES = PARTIAL_GRAD_PREFIXES + ("patch_linear.", "processor.")


class SeqAttnFn(torch.autograd.Function):
    """The attention layer of the sequence-parallel step in bf16 mode, exchange included (functional.MultiLinearFn +
    SeqToHeadsFn + AttentionFn + HeadsToSeqFn of the fp32 form as ONE node):

      forward : q|k|v projection of MY rows written by the GEMM epilogue straight as the all-to-all's bf16 send buffer
                (RoPE at the rows' global positions, q scale folded in: csrc/gemm_k256.hip, gaot_qkv_image_packed) -> all-to-all
                -> what arrives IS my heads' attention image -> flash kernels -> O rounded to bf16 -> all-to-all -> my rows,
                all heads, widened back to fp32 for o_proj (which rounds its A operand to bf16 anyway: no extra rounding)
      backward: dO of my rows packed + rounded to bf16 in one pass -> all-to-all -> it IS the kernels' dO image (delta from it)
                -> dK/dV, dQ -> d(q|k|v) rounded to bf16 -> all-to-all -> unpacked as the bf16 A operand of the two GEMMs
                dx = d W and dW = d^T x.
    Every exchanged tensor is bf16 (half the bytes of the fp32 form) and the only glue kernels are the two casts and three
    pack / unpack launches; the roundings are the ones the consuming MFMA kernels apply to these operands in any case."""

    @staticmethod
    def forward(ctx, x: Tensor, group, h: int, hkv: int, freqs: Optional[Tensor], dropout_p: float, wq: Tensor, wk: Tensor,
                wv: Tensor):
        from . import functional as GF
        from . import ops
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        r, k = x.shape[-2], x.shape[-1]
        xb = GF.bf16_copy_of(x, (r, k))
        ws = [GF._w2d(w) for w in (wq, wk, wv)]
        ntot = (h + 2 * hkv) * 32
        wcat = GF._wb(ws[0].new_empty(0).set_(ws[0].untyped_storage(), ws[0].storage_offset(), (ntot, k), (k, 1)), 1)
        s_total = r * world
        scale = 1.0 / (32 ** 0.5)
        hl, kl = h // world, hkv // world
        lw = (hl + 2 * kl) * 32
        send = ops.qkv_image_packed(xb, wcat, r, rank * r, s_total, h, hkv, freqs, scale, world)
        img = ops._ws(_lib_image_bytes(1, s_total, hl, kl), x.device)                 # image + the kernels' scratch behind it
        recv = img[:s_total * lw * 2].view(torch.bfloat16).view(world, r, lw)
        _all_to_all_into(recv, send, group)
        # ONE seed word for all ranks (the common stream); the kernels key a head's mask by its GLOBAL index, so the rank that
        # owns a head draws the mask the unsharded step draws for it
        seed = GF.next_dropout_seed(x.device) if dropout_p > 0.0 else None
        o, lse, _ = ops.attn_fwd_bf16(None, None, 1, s_total, hl, kl, scale, dropout_p, seed, image=img, head0=rank * hl,
                                      heads_total=h)
        ob = ops.cast_bf16(o)                                                         # [S, hl*32] = [world, r, hl*32] blocks
        orecv = torch.empty(world, r, hl * 32, dtype=torch.bfloat16, device=x.device)
        _all_to_all_into(orecv, ob.view(world, r, hl * 32), group)
        out = torch.empty(r, h * 32, dtype=torch.float32, device=x.device)
        ops.pack_heads(out, orecv, world, [(0, hl * 32)], False)
        empty = torch.empty(0, device=x.device)
        ctx.save_for_backward(xb, wcat, img, o, lse, freqs if freqs is not None else empty, seed if seed is not None else empty)
        ctx.meta = (group, world, rank, h, hkv, r, k, scale, dropout_p, x.shape, [w.shape for w in (wq, wk, wv)])
        return out.view(*x.shape[:-1], h * 32)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        from . import functional as GF
        from . import ops
        xb, wcat, img, o, lse, freqs, seed = ctx.saved_tensors
        group, world, rank, h, hkv, r, k, scale, dropout_p, xshape, wshapes = ctx.meta
        hl, kl = h // world, hkv // world
        lw = (hl + 2 * kl) * 32
        s_total = r * world
        d = d_out.reshape(r, h * 32)
        d = d if d.is_contiguous() else d.contiguous()
        dsend = torch.empty(world, r, hl * 32, dtype=torch.bfloat16, device=d.device)
        ops.pack_heads(d, dsend, world, [(0, hl * 32)], True)                         # fp32 rows -> bf16 blocks, one pass
        scratch = ops.attn_bwd_scratch(1, s_total, hl, kl, d.device)
        dorecv = scratch[:s_total * hl * 32 * 2].view(torch.bfloat16).view(world, r, hl * 32)
        _all_to_all_into(dorecv, dsend, group)                                        # = the kernels' dO image of my heads
        dqkv = ops.attn_bwd_bf16(img, o, None, lse, 1, s_total, hl, kl, scale, dropout_p, seed if dropout_p > 0.0 else None,
                                 freqs if freqs.numel() else None, do_image=scratch, head0=rank * hl, heads_total=h)
        gsend = ops.cast_bf16(dqkv).view(world, r, lw)
        grecv = torch.empty(world, r, lw, dtype=torch.bfloat16, device=d.device)
        _all_to_all_into(grecv, gsend, group)
        ntot = (h + 2 * hkv) * 32
        dproj = torch.empty(r, ntot, dtype=torch.bfloat16, device=d.device)           # d(q|k|v) of my rows, fused layout
        ops.pack_heads(dproj, grecv, world, qkv_segments(h, hkv, world), False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dproj, wcat, r, k, ntot, ntot, k, False, False, precision=1).view(xshape)
        dwcat = GF._dw_gemm(dproj, xb, ntot, k, r, ntot, k, 1)
        dws, col = [], 0
        for shp in wshapes:
            dws.append(dwcat[col:col + shp[0]].view(shp))
            col += shp[0]
        return (dx, None, None, None, None, None, *dws)


def _lib_image_bytes(b: int, s: int, h: int, hkv: int) -> int:
    from . import _lib
    return int(_lib.load().gaot_attn_bf16_image_bytes(b, s, h, hkv))


def seq_attn_eligible(x: Tensor, module) -> bool:
    """the bf16 sequence-parallel attention node applies: bf16 mode, head_dim 32, d_model 256, the producer's bf16 image"""
    from . import functional as GF
    from . import ops
    return bool(ops.get_precision() == "bf16" and x.is_cuda and module.head_dim == 32 and x.shape[-1] == 256
                and GF.bf16_copy_of(x, (x.shape[-2], x.shape[-1])) is not None
                and ((module.num_heads + 2 * module.num_kv_heads) * 32) % 64 == 0
                and all(w.requires_grad for w in (module.q_proj.weight, module.k_proj.weight, module.v_proj.weight)))


def partial_grad_parameters(model, parallel: str = "head") -> List[torch.nn.Parameter]:
    pre = SEQ_PARTIAL_GRAD_PREFIXES if parallel == "seq" else PARTIAL_GRAD_PREFIXES
    return [p for k, p in model.named_parameters() if p.requires_grad and k.startswith(pre)]


def allreduce_partial_grads(params: List[torch.nn.Parameter], group):
    """one flat SUM all-reduce over the gradients of the per-point / per-edge parameters.  A parameter without a local
    gradient (a rank whose shard holds no edge of that operator) contributes zeros, so the flat buffer has the same size on
    every rank, and every rank ends up with the summed gradient in ``p.grad``."""
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    comm.run(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group), (flat,), "all_reduce")
    for p, v in zip(params, flat.split([p.numel() for p in params])):
        if p.grad is None:
            p.grad = v.view_as(p).clone()
        else:
            p.grad.copy_(v.view_as(p))


class GradBuckets:
    """SUM all-reduce of the partial parameter gradients in buckets, each launched (asynchronously, on its own process
    group = its own RCCL stream) as soon as backward has produced the last gradient of the bucket AND every earlier bucket is
    launched, so that the exchange of the 45 MB of Transformer weight gradients of the sequence-parallel step runs under
    the remaining backward instead of after it.  Buckets follow the order in which backward finishes parameters (reverse
    registration order) and are issued STRICTLY in index order on every rank: a rank whose shard produced no gradient for
    one parameter of bucket i launches it (and every later one) in ``finish()`` while the other ranks launch it from the
    hooks -- the sequence of collectives on the group is the same everywhere, only the time differs.  After
    ``finish()`` every ``p.grad`` is a view into its bucket's flat buffer holding the summed gradient."""

    def __init__(self, params: List[torch.nn.Parameter], group, bucket_bytes: int = 12 << 20):
        self.group = group
        self.buckets = []          # [params, flat, views, pending, handle, launched]
        order = list(reversed(params))
        cur, size = [], 0
        for p in order:
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._add(cur)
                cur, size = [], 0
        if cur:
            self._add(cur)
        self._index = {}
        self._hooks = []
        self._next = 0             # first bucket not launched yet: the only one a hook may launch
        self._unused = None        # ids of parameters NO rank produces a gradient for (refreshed at every eager finish())
        self._late = set()         # ids of such parameters whose first gradient arrived behind their bucket's launch (this step)
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[id(p)] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _add(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.buckets.append(dict(params=list(ps), flat=flat, views=views, pending=len(ps), handle=None, launched=False))

    def reset(self):
        """start of a step: no gradient seen yet, nothing in flight (``forward_backward`` calls it; an exception in the
        middle of a backward leaves stale counters otherwise)"""
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None
            b["pending"], b["launched"] = self._expected(b), False
        self._next = 0

    def _expected(self, b):
        """gradients a bucket waits for: its parameters minus those NO rank produces a gradient for (the same set on every
        rank: it comes out of an all-reduce), so that an unused parameter (``skip_proj`` without long-range skips) does not
        keep its bucket and every later one out of the backward pass"""
        unused = self._unused or ()
        return sum(1 for p in b["params"] if id(p) not in unused)

    def _on_grad(self, p):
        b = self.buckets[self._index[id(p)]]
        if id(p) in (self._unused or ()):
            # a parameter no rank had a gradient for so far starts to receive one.  Its bucket does not wait for it: if the
            # bucket has not left yet, _launch takes the gradient along; if it is in flight already, the gradient stays where
            # autograd put it and the eager finish() of this step adds it to the bucket's sum on every rank (_find_unused).
            # Either way the parameter counts towards its bucket from the next step on
            if b["launched"]:
                self._late.add(id(p))
            return
        if b["launched"]:
            # a second backward before finish() (gradient accumulation) would accumulate in place into the flat buffer
            # while its all-reduce may still be in flight on the other communicator: refuse instead of racing
            raise RuntimeError("GradBuckets: a gradient arrived for a bucket whose all-reduce is already launched; call "
                               "finish() (or reset()) between two backward passes")
        b["pending"] -= 1
        # index order: a complete bucket behind an incomplete one waits for it (finish() at the latest)
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] <= 0:
            self._launch(self.buckets[self._next])

    def _launch(self, b):
        # EVERY parameter of the bucket ends up with p.grad = its view of the flat buffer -- also one that received no
        # gradient on this rank (a shard without an edge of that operator): its view is zero-filled here and holds the other
        # ranks' sum afterwards, so the optimizer steps it on every rank alike and the replicas cannot drift apart.  A
        # parameter NO rank has a gradient for keeps grad = None (``_unused``), its view is zeroed all the same so that a
        # stale sum of an earlier step can never be added again
        assert b is self.buckets[self._next]
        copy_dst, copy_src = [], []
        unused = self._unused or ()
        for p, v in zip(b["params"], b["views"]):
            g = p.grad
            if g is None:
                v.zero_()
                if id(p) in unused:
                    continue           # e.g. skip_proj without long-range skips: stays None, the optimizer skips it
            elif g.data_ptr() != v.data_ptr():
                copy_dst.append(v)
                copy_src.append(g)
            p.grad = v
        # with weight gradients on a side stream (overlap_dw) the copies into the flat buffer and the all-reduce's dependency
        # follow them there: the bucket leaves when ITS gradients are done, and the step's stream never waits for it
        if copy_dst:
            comm.side_run(lambda: torch._foreach_copy_(copy_dst, copy_src), tuple(copy_dst) + tuple(copy_src))
        flat, group = b["flat"], self.group
        side = comm.side_stream()          # bound now: a replayed closure runs when the module-level setting is long reset

        def issue():
            if side is None:
                b["handle"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
            else:
                side.wait_stream(torch.cuda.current_stream())      # gradients the chain itself produced (biases, norms)
                with torch.cuda.stream(side):
                    b["handle"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        comm.run(issue, (flat,), "grad_bucket_issue")
        b["launched"] = True
        self._next += 1

    def finish(self):
        """launch what backward left incomplete (parameters that received no gradient) in index order, then wait for every
        bucket; after it every ``p.grad`` of every bucket is a view into the flat buffer holding the sum over the ranks"""
        eager = not comm.capturing()
        if not eager and self._unused is None:
            raise RuntimeError("GradBuckets: run one eager step before capturing (which parameters receive a gradient on "
                               "no rank is found by an all-reduce that needs a host read)")
        # which parameters have a LOCAL gradient this step (a launched bucket has re-pointed its gradients: read before)
        had = None
        if eager:
            had = self._local_flags()
        while self._next < len(self.buckets):
            self._launch(self.buckets[self._next])
        if eager:
            self._find_unused(had)         # AFTER the last bucket: the same position in every rank's sequence
        self._late.clear()
        buckets = self.buckets

        def wait():
            for b in buckets:
                if b["handle"] is not None:
                    b["handle"].wait()
                    b["handle"] = None
        comm.run(wait, (), "grad_bucket_wait")
        for b in self.buckets:
            b["pending"], b["launched"] = self._expected(b), False
        self._next = 0

    def _local_flags(self):
        """1.0 per parameter that holds a gradient on THIS rank after this backward, read before ``finish()`` launches the
        rest: a bucket launched from the hooks had every expected gradient (its ``p.grad`` are views now, an unused
        parameter's is still None), the others are as backward left them.  A gradient left over from an earlier step counts
        exactly as an unsharded ``p.grad`` left over would (callers clear gradients between steps)."""
        return [0.0 if p.grad is None else 1.0 for b in self.buckets for p in b["params"]]

    def _find_unused(self, had):
        """one tiny all-reduce per eager step, issued behind the last bucket on every rank: which parameters got a gradient
        on NO rank.  Those keep ``grad = None`` as in an unsharded run (the optimizer skips them: reference semantics for
        ``skip_proj`` with ``use_long_range_skip=False``, attn.py:321); a parameter that only THIS rank has no gradient for
        was zero-filled and now holds the other ranks' sum.  Refreshed every eager step (every rank must take part, so no rank
        can skip it on its own; the steady state is graph replay, where the set is frozen), so a parameter that starts to
        receive gradients later is picked up by every rank in the same step: its gradient is summed over the ranks here,
        outside its bucket, and it counts towards its bucket from the next step on."""
        ps = [p for b in self.buckets for p in b["params"]]
        late = [1.0 if id(p) in self._late else 0.0 for p in ps]
        flags = torch.tensor(list(had) + late, dtype=torch.float32, device=ps[0].device)
        dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=self.group)
        flags = flags.tolist()
        unused = {id(p) for p, f in zip(ps, flags[:len(ps)]) if f == 0.0}
        late_any = {id(p) for p, f in zip(ps, flags[len(ps):]) if f > 0.0}
        for b in self.buckets:
            for p, v in zip(b["params"], b["views"]):
                if id(p) in late_any:
                    # on some rank the first gradient of this parameter arrived behind its bucket's launch (the same decision
                    # everywhere: it comes out of the all-reduce above).  The bucket's sum holds the ranks that were in time;
                    # the late ranks' gradients are summed here and added to it, on every rank alike
                    if b["handle"] is not None:
                        b["handle"].wait()
                        b["handle"] = None
                    mine = p.grad if (id(p) in self._late and p.grad is not None) else torch.zeros_like(v)
                    extra = mine.detach().clone()
                    dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group)
                    v.add_(extra)
                    p.grad = v
                elif id(p) in unused:
                    p.grad = None          # (a zero-filled view while the set was not known yet)
                elif p.grad is None:
                    p.grad = v             # left None by an older set: the view holds the other ranks' sum
        self._unused = unused

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class ShardedStep:
    """forward + MSE + backward of the drop-in model on a rank-local shard (see module docstring)."""

    def __init__(self, model, group, n_total: int, head_parallel: bool = True, parallel: Optional[str] = None,
                 grad_group=None, overlap_grads: bool = True, overlap_dw: bool = False):
        """``parallel``: how the latent Transformer is divided -- "seq" (token rows per rank, heads per rank inside
        attention), "head" (replicated except the attention heads), "replicated"; default "seq" when heads and token rows
        divide over the ranks, else "head" (``head_parallel=False`` -> "replicated").
        ``grad_group``: a second process group over the same ranks for the bucketed weight-gradient all-reduce (its own RCCL
        stream, so the buckets overlap the exchange steps of the remaining backward); default: ``group``.
        ``overlap_grads=False``: one flat all-reduce after backward instead of buckets launched from gradient hooks.
        ``overlap_dw=True``: the weight-gradient GEMMs (and the bucket copies / all-reduces that follow them) run on a side
        stream beside the backward chain and its exchange steps, joined once before the step returns (``comm.side_run``);
        measured SLOWER on one GPU (no idle CUs beside the chain: profiles/archive/r4_u), meant for N > 1 where the chain waits in
        exchange steps -- ``bench.py --gpus N`` times both settings."""
        self.model = model
        self.group = group
        self.n_total = n_total
        world = dist.get_world_size(group)
        attn = [m for m in model.modules() if hasattr(m, "num_kv_heads") and hasattr(m, "o_proj")]
        divisible = all(m.num_heads % world == 0 and m.num_kv_heads % world == 0 for m in attn)
        s_tok = model.num_latent_tokens // model.patch_size ** 3
        if parallel is None:
            parallel = "replicated" if not head_parallel else ("seq" if (divisible and s_tok % world == 0) else "head")
        if parallel not in ("seq", "head", "replicated"):
            raise ValueError(f"parallel must be 'seq', 'head' or 'replicated', got {parallel}")
        if parallel == "seq" and not (divisible and s_tok % world == 0):
            raise ValueError(f"parallel='seq' needs heads and {s_tok} token rows to divide over {world} ranks")
        self.parallel = parallel
        self.partial = partial_grad_parameters(model, parallel)
        self.buckets = GradBuckets(self.partial, grad_group if grad_group is not None else group) if overlap_grads else None
        self.overlap_dw = bool(overlap_dw)
        self._side = torch.cuda.Stream() if (self.overlap_dw and torch.cuda.is_available()) else None
        model.encoder._shard_group = group
        model.decoder._shard_group = group
        model._shard_group = group
        model._seq_group = group if parallel == "seq" else None
        for mod in attn:
            # "head": the Transformer is replicated, every rank holds the same q|k|v, computes its share of the heads and
            # the outputs / gradients are all-gathered; "seq": all-to-all around the kernels (SeqAttnFn / AttentionFn)
            mod._head_group = group if parallel == "head" else None
            mod._seq_group = group if parallel == "seq" else None

    def set_overlap_dw(self, on: bool) -> None:
        """switch the weight-gradient side stream between steps (a recorded SegmentedGraph keeps the setting it was
        recorded with)"""
        self.overlap_dw = bool(on)
        if self.overlap_dw and self._side is None:
            self._side = torch.cuda.Stream()

    def release(self):
        """undo the hooks on the model (tests)"""
        m = self.model
        m.encoder._shard_group = m.decoder._shard_group = m._shard_group = m._seq_group = None
        for mod in m.modules():
            if hasattr(mod, "_head_group"):
                mod._head_group = None
                mod._seq_group = None
        if self.buckets is not None:
            self.buckets.remove()

    def forward_backward(self, batch: MeshBatch, tokens_pos: Optional[Tensor]):
        from . import functional as GF
        if self.buckets is not None:
            self.buckets.reset()
        pred = self.model(batch=batch, tokens_pos=tokens_pos)
        n_local = pred.shape[0]
        # global MSE = sum over ranks of (local sum of squares) / (N_total * out)
        loss = GF.mse_loss(pred, batch.x) * (float(n_local) / float(self.n_total))
        comm.set_side_stream(self._side if self.overlap_dw else None)
        try:
            loss.backward()
            if self.buckets is not None:
                self.buckets.finish()
            comm.side_join()               # every weight gradient is complete on the step's stream from here on
        finally:
            comm.set_side_stream(None)
        if self.buckets is None:
            allreduce_partial_grads(self.partial, self.group)
        total = loss.detach().clone()
        group = self.group
        comm.run(lambda: dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group), (total,), "all_reduce")
        return total
