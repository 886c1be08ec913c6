"""Host-side companions of the training step that the reference keeps in its trainer: the 'mix'
learning-rate schedule (src/trainer/optimizers.py:40-67, 226-246) for gaot_3d_amd.optim.AdamW, and the random node
sub-sampling of the neural-field training strategy (src/trainer/stat.py:438-514) done on the device."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch.optim.lr_scheduler import _LRScheduler

from .data import MeshBatch


def mix_phases(total_epochs: int) -> Tuple[int, int, int]:
    """(warm-up, cosine, exponential) epoch counts: 2 % / 90 % / rest, each at least 1 (optimizers.py:226-235)"""
    warm = int(0.02 * total_epochs)
    cos = int(0.90 * total_epochs)
    exp = total_epochs - warm - cos
    if warm == 0:
        warm, cos = 1, cos - 1
    if exp == 0:
        exp, cos = 1, cos - 1
    return warm, cos, exp


def mix_lr(epoch: int, warm: int, cos: int, exp: int, initial_lr: float, max_lr: float, min_lr: float, final_lr: float) -> float:
    """learning rate of scheduler step ``epoch`` (optimizers.py:53-67): linear initial->max, cosine max->min, then
    exponential min->final"""
    if epoch < warm:
        return initial_lr + (max_lr - initial_lr) * (epoch / max(1, warm - 1))
    if epoch < warm + cos:
        ratio = (1 + math.cos(math.pi * (epoch - warm) / cos)) / 2
        return min_lr + (max_lr - min_lr) * ratio
    return min_lr * ((final_lr / min_lr) ** ((epoch - warm - cos) / max(1, exp - 1)))


class MixLRScheduler(_LRScheduler):
    """the reference's CustomLRScheduler, with the phase split of AdamWOptimizer built in"""

    def __init__(self, optimizer, total_epochs: int, initial_lr: float, max_lr: float, min_lr: float, final_lr: float,
                 last_epoch: int = -1):
        self.phases = mix_phases(total_epochs)
        self.lrs = (initial_lr, max_lr, min_lr, final_lr)
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        lr = mix_lr(self.last_epoch, *self.phases, *self.lrs)
        return [lr for _ in self.optimizer.param_groups]


def sample_nodes_neural_field(batch: MeshBatch, num_input_nodes: int, num_query_nodes: int,
                              generator: Optional[torch.Generator] = None):
    """stat.py:438-514 on the batch's own device: per graph a random subset (without replacement) of
    ``num_input_nodes`` points as encoder input and of ``num_query_nodes`` points as decoder queries (the SAME subset
    when the two counts are equal).  Returns (sampled_batch, query_pos, query_batch_idx, target_for_loss); the sampled
    batch carries pos / x / c only -- the graphs of a neural-field step are built by the model (precompute_edges=False)."""
    dev = batch.pos.device
    ptr = batch.ptr.tolist() if hasattr(batch, "ptr") and batch.ptr is not None else [0, batch.pos.shape[0]]
    same = num_input_nodes == num_query_nodes
    in_idx, q_idx, q_b, sizes = [], [], [], []
    for i in range(batch.num_graphs):
        lo, hi = ptr[i], ptr[i + 1]
        n = hi - lo
        n_in = min(num_input_nodes, n)
        if n_in <= 0:
            continue
        ip = torch.randperm(n, device=dev, generator=generator)[:n_in] + lo
        qp = ip if same else torch.randperm(n, device=dev, generator=generator)[:min(num_query_nodes, n)] + lo
        in_idx.append(ip)
        q_idx.append(qp)
        q_b.append(torch.full((qp.numel(),), len(sizes), dtype=torch.long, device=dev))
        sizes.append(n_in)
    ii, qi = torch.cat(in_idx), torch.cat(q_idx)
    out = MeshBatch(pos=batch.pos[ii], x=batch.x[ii])
    if getattr(batch, "c", None) is not None:
        out.c = batch.c[ii]
    out.num_graphs = len(sizes)
    out.batch = torch.repeat_interleave(torch.arange(len(sizes), device=dev), torch.tensor(sizes, device=dev))
    out.ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.long, device=dev)
    for attr in ("filename", "num_latent_nodes"):
        if hasattr(batch, attr):
            setattr(out, attr, getattr(batch, attr))
    return out, batch.pos[qi], torch.cat(q_b), batch.x[qi]
