#!/usr/bin/env python3
"""Which Python lines of one eager training step at configs[1] launch ATen kernels (fills, copies, adds)?  Prints the
aten ops of one step grouped by the innermost gaot_3d_amd / bench frame.  usage: find_aten_kernels.py [points]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaot_3d_amd
from gaot_3d_amd import functional as GF
from gaot_3d_amd.data import make_synthetic_sample
from gaot_3d_amd.model import init_model
from gaot_3d_amd.optim import AdamW
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda:0")
gaot_3d_amd.set_precision("bf16")
cfg = bench.model_config((64, 64, 32), 10, 8, 0.1)
torch.manual_seed(0)
model = init_model(6, 1, "gaot_3d", cfg).to(dev).train()
opt = AdamW(model.parameters(), lr=3e-4, weight_decay=1e-5)
batch, tokens = make_synthetic_sample(n, (64, 64, 32), k=8, seed=0, device=str(dev))
tokens = tokens.to(dev)


def step():
    gaot_3d_amd.clear_graph_cache(batch)
    opt.zero_grad(set_to_none=True)
    pred = model(batch=batch, tokens_pos=tokens)
    loss = GF.mse_loss(pred, batch.x)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
evs = list(prof.events())
# parent chain: the innermost enclosing non-aten event (autograd node / Function name) tells who asked for the op
for ev in evs:
    if not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::add", "aten::add_", "aten::clone",
                       "aten::contiguous", "aten::cat", "aten::mul", "aten::ones_like", "aten::full"):
        continue
    par = ev.cpu_parent
    chain = []
    while par is not None and len(chain) < 4:
        chain.append(par.name)
        par = par.cpu_parent
    where = ""
    for fr in (ev.stack or []):
        if "gaot_3d_amd" in fr or "find_aten" in fr:
            where = fr.strip()[-80:]
            break
    shape = ""
    cnt[(ev.name, " < ".join(chain)[:120], where)] += 1
for (name, chain, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:4d}  {name:18s} {chain}  {where}")
