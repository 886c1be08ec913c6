"""Point-sharding of ONE sample across the GPUs of a node (new w.r.t. the reference, whose only
parallelism is sample-level DDP, src/trainer/stat.py:431-436).

Partition: physical points are split into `world` contiguous ranges; every edge lives with its PHYSICAL
endpoint (encoder: source, decoder: query), so edges, features, targets and predictions are disjoint across
ranks.  Latent coordinates, the latent Transformer and all parameters are replicated.

Exchange steps (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests):
  forward : encoder GNO   per-token SUM [M,C] and COUNT [M]  -> all_reduce(SUM) -> mean = sum / max(count,1)
  backward: decoder       partial d(loss)/d(latent) [M,C]    -> all_reduce(SUM) before the (replicated, hence
            identical) Transformer backward
  after backward: gradients of the per-point / per-edge parameters (encoder.lifting, encoder.gno, decoder.*)
            are partial sums -> one flat all_reduce (~125 KB).  Per-token parameters (encoder.geoembed,
            encoder.recovery, patch_linear, processor.*) see identical full gradients on every rank.
Head-parallel attention (on by default in ShardedStep): rank r runs the attention kernels for heads [r*H/G, (r+1)*H/G)
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
        s = local_mean * local_deg[:, None]
        d = local_deg.clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
        dg = d.clamp(min=1.0)
        ctx.save_for_backward(local_deg / dg)
        return s / dg[:, None]

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


PARTIAL_GRAD_PREFIXES = ("encoder.lifting.", "encoder.gno.", "decoder.")


def partial_grad_parameters(model) -> List[torch.nn.Parameter]:
    return [p for k, p in model.named_parameters() if p.requires_grad and k.startswith(PARTIAL_GRAD_PREFIXES)]


def allreduce_partial_grads(params: List[torch.nn.Parameter], group):
    """one flat SUM all-reduce over the gradients of the per-point / per-edge parameters"""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class ShardedStep:
    """forward + MSE + backward of the drop-in model on a rank-local shard (see module docstring)."""

    def __init__(self, model, group, n_total: int, head_parallel: bool = True):
        self.model = model
        self.group = group
        self.n_total = n_total
        self.partial = partial_grad_parameters(model)
        model.encoder._shard_group = group
        model.decoder._shard_group = group
        model._shard_group = group
        # head-parallel attention: the Transformer is replicated, so every rank holds the same q|k|v; each
        # computes its share of the heads and the outputs / gradients are all-gathered (functional.AttentionFn)
        for mod in model.modules():
            if hasattr(mod, "num_kv_heads") and hasattr(mod, "o_proj"):
                mod._head_group = group if head_parallel else None

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
