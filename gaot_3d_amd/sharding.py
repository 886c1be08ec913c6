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
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        dg = buf[:, -1].clamp(min=1.0)
        ctx.save_for_backward(local_deg / dg)
        return buf[:, :-1] / dg[:, None]

    @staticmethod
    def backward(ctx, g: Tensor):
        (ratio,) = ctx.saved_tensors
        return g * ratio[:, None], None, None


class AllReduceGradFn(torch.autograd.Function):
    """identity in forward; SUM all-reduce of the gradient in backward (partial -> full)."""

    @staticmethod
    def forward(ctx, x: Tensor, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


# ---- head-parallel attention: the index arithmetic of the exchange (device independent; kernels live in functional.py)
def head_slices(rank: int, world: int, h: int, hkv: int):
    """column ranges of rank's q / k / v heads inside a fused [rows, (h + 2 hkv) * 32] projection"""
    hl, kl = h // world, hkv // world
    return ((rank * hl * 32, (rank + 1) * hl * 32), ((h + rank * kl) * 32, (h + (rank + 1) * kl) * 32),
            ((h + hkv + rank * kl) * 32, (h + hkv + (rank + 1) * kl) * 32))


def local_qkv(qkv: Tensor, rank: int, world: int, h: int, hkv: int) -> Tensor:
    (q0, q1), (k0, k1), (v0, v1) = head_slices(rank, world, h, hkv)
    return torch.cat([qkv[:, q0:q1], qkv[:, k0:k1], qkv[:, v0:v1]], dim=1)


def all_gather_stack(t: Tensor, group, world: int) -> Tensor:
    """[world, *t.shape]: one all-gather (RCCL: all_gather_into_tensor; gloo, used by the tests: list form)"""
    t = t if t.is_contiguous() else t.contiguous()
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return torch.stack(parts)


def gather_head_outputs(o_local: Tensor, group, world: int) -> Tensor:
    """[rows, (h/world)*32] per rank -> [rows, h*32] with the heads back in global order"""
    allo = all_gather_stack(o_local, group, world)
    return allo.permute(1, 0, 2).reshape(o_local.shape[0], world * o_local.shape[1])


def gather_qkv_grads(dqkv_local: Tensor, group, world: int, h: int, hkv: int) -> Tensor:
    """[rows, (hl + 2 kl)*32] per rank (q | k | v of the rank's heads) -> [rows, (h + 2 hkv)*32] in the fused layout"""
    hl, kl = h // world, hkv // world
    rows = dqkv_local.shape[0]
    allg = all_gather_stack(dqkv_local, group, world).permute(1, 0, 2)           # [rows, world, (hl + 2 kl) * 32]
    return torch.cat([allg[:, :, :hl * 32].reshape(rows, h * 32),
                      allg[:, :, hl * 32:(hl + kl) * 32].reshape(rows, hkv * 32),
                      allg[:, :, (hl + kl) * 32:].reshape(rows, hkv * 32)], dim=1)


# ---- sequence-parallel Transformer: the exchange steps as autograd Functions (device independent) ----------------------
def _all_to_all(send: Tensor, group) -> Tensor:
    """block j of ``send`` [G, ...] goes to rank j; block i of the result came from rank i.  RCCL: all_to_all_single; the
    gloo group of the one-GPU tests cannot exchange device tensors this way, so they are staged through the host there"""
    send = send if send.is_contiguous() else send.contiguous()
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        h = send.cpu()
        r = torch.empty_like(h)
        dist.all_to_all_single(r, h, group=group)
        return r.to(send.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv


def _reduce_scatter_rows(t: Tensor, group, world: int, rank: int) -> Tensor:
    """sum over the ranks of ``t`` [world * r, ...], rank keeps its r rows (RCCL: reduce_scatter_tensor)"""
    t = t if t.is_contiguous() else t.contiguous()
    r = t.shape[0] // world
    if dist.get_backend(group) == "nccl":
        out = torch.empty((r,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=group)
        return out
    full = t.clone()
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
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


def _pack_heads(qkv: Tensor, world: int, h: int, hkv: int) -> Tensor:
    """fused [rows, (h + 2 hkv) * 32] -> [world, rows, (hl + 2 kl) * 32]: block j = q | k | v of rank j's heads"""
    rows = qkv.shape[0]
    hl, kl = h // world * 32, hkv // world * 32
    out = torch.empty(world, rows, hl + 2 * kl, dtype=qkv.dtype, device=qkv.device)
    out[:, :, :hl] = qkv[:, :h * 32].view(rows, world, hl).permute(1, 0, 2)
    out[:, :, hl:hl + kl] = qkv[:, h * 32:(h + hkv) * 32].view(rows, world, kl).permute(1, 0, 2)
    out[:, :, hl + kl:] = qkv[:, (h + hkv) * 32:].view(rows, world, kl).permute(1, 0, 2)
    return out


def _unpack_heads(blocks: Tensor, world: int, h: int, hkv: int) -> Tensor:
    """inverse of _pack_heads"""
    rows = blocks.shape[1]
    hl, kl = h // world * 32, hkv // world * 32
    out = torch.empty(rows, (h + 2 * hkv) * 32, dtype=blocks.dtype, device=blocks.device)
    out[:, :h * 32].view(rows, world, hl).copy_(blocks[:, :, :hl].permute(1, 0, 2))
    out[:, h * 32:(h + hkv) * 32].view(rows, world, kl).copy_(blocks[:, :, hl:hl + kl].permute(1, 0, 2))
    out[:, (h + hkv) * 32:].view(rows, world, kl).copy_(blocks[:, :, hl + kl:].permute(1, 0, 2))
    return out


class SeqToHeadsFn(torch.autograd.Function):
    """q|k|v of MY token rows, all heads [S/G, (h + 2 hkv) * 32]  ->  q|k|v of ALL token rows, my heads
    [S, (h/G + 2 hkv/G) * 32] (one all-to-all); backward is the inverse exchange of the gradient"""

    @staticmethod
    def forward(ctx, qkv: Tensor, group, h: int, hkv: int):
        world = dist.get_world_size(group)
        if h % world or hkv % world:
            raise ValueError(f"sequence-parallel attention: {h} / {hkv} heads do not divide over {world} ranks")
        ctx.meta = (group, world, h, hkv)
        recv = _all_to_all(_pack_heads(qkv, world, h, hkv), group)          # block i = rank i's rows
        return recv.view(world * qkv.shape[0], recv.shape[2])

    @staticmethod
    def backward(ctx, g: Tensor):
        group, world, h, hkv = ctx.meta
        g = g if g.is_contiguous() else g.contiguous()
        recv = _all_to_all(g.view(world, g.shape[0] // world, g.shape[1]), group)   # block j = my rows, rank j's heads
        return _unpack_heads(recv, world, h, hkv), None, None, None


class HeadsToSeqFn(torch.autograd.Function):
    """attention output of ALL rows, my heads [S, (h/G) * 32] -> MY rows, all heads [S/G, h * 32] (heads in global order)"""

    @staticmethod
    def forward(ctx, o: Tensor, group):
        world = dist.get_world_size(group)
        ctx.meta = (group, world)
        o = o if o.is_contiguous() else o.contiguous()
        rows = o.shape[0] // world
        recv = _all_to_all(o.view(world, rows, o.shape[1]), group)          # block j = my rows, rank j's heads
        return recv.permute(1, 0, 2).reshape(rows, world * o.shape[1])

    @staticmethod
    def backward(ctx, g: Tensor):
        group, world = ctx.meta
        rows, hl = g.shape[0], g.shape[1] // world
        recv = _all_to_all(g.view(rows, world, hl).permute(1, 0, 2).contiguous(), group)   # block i = rank i's rows, my heads
        return recv.view(world * rows, hl), None


PARTIAL_GRAD_PREFIXES = ("encoder.lifting.", "encoder.gno.", "decoder.")
SEQ_PARTIAL_GRAD_PREFIXES = PARTIAL_GRAD_PREFIXES + ("patch_linear.", "processor.")


def partial_grad_parameters(model, parallel: str = "head") -> List[torch.nn.Parameter]:
    pre = SEQ_PARTIAL_GRAD_PREFIXES if parallel == "seq" else PARTIAL_GRAD_PREFIXES
    return [p for k, p in model.named_parameters() if p.requires_grad and k.startswith(pre)]


def allreduce_partial_grads(params: List[torch.nn.Parameter], group):
    """one flat SUM all-reduce over the gradients of the per-point / per-edge parameters"""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])


class ShardedStep:
    """forward + MSE + backward of the drop-in model on a rank-local shard (see module docstring)."""

    def __init__(self, model, group, n_total: int, head_parallel: bool = True, parallel: Optional[str] = None):
        """``parallel``: how the latent Transformer is divided -- "seq" (token rows per rank, heads per rank inside
        attention), "head" (replicated except the attention heads), "replicated"; default "seq" when heads and token rows
        divide over the ranks, else "head" (``head_parallel=False`` -> "replicated")"""
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
        model.encoder._shard_group = group
        model.decoder._shard_group = group
        model._shard_group = group
        model._seq_group = group if parallel == "seq" else None
        for mod in attn:
            # "head": the Transformer is replicated, every rank holds the same q|k|v, computes its share of the heads and
            # the outputs / gradients are all-gathered; "seq": all-to-all around the kernels (functional.AttentionFn)
            mod._head_group = group if parallel == "head" else None
            mod._seq_group = group if parallel == "seq" else None

    def release(self):
        """undo the hooks on the model (tests)"""
        m = self.model
        m.encoder._shard_group = m.decoder._shard_group = m._shard_group = m._seq_group = None
        for mod in m.modules():
            if hasattr(mod, "_head_group"):
                mod._head_group = None
                mod._seq_group = None

    def forward_backward(self, batch: MeshBatch, tokens_pos: Optional[Tensor]):
        from . import functional as GF
        pred = self.model(batch=batch, tokens_pos=tokens_pos)
        n_local = pred.shape[0]
        # global MSE = sum over ranks of (local sum of squares) / (N_total * out)
        loss = GF.mse_loss(pred, batch.x) * (float(n_local) / float(self.n_total))
        loss.backward()
        allreduce_partial_grads(self.partial, self.group)
        total = loss.detach().clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return total
