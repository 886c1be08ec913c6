#!/usr/bin/env python3
"""cProfile of the host side of eager training steps at configs[1] (where do the ~15 ms of Python per step go?)"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaot_3d_amd
from gaot_3d_amd import functional as GF
from gaot_3d_amd.data import make_synthetic_sample
from gaot_3d_amd.model import init_model
from gaot_3d_amd.optim import AdamW
import bench

dev = torch.device("cuda:0")
gaot_3d_amd.set_precision("bf16")
cfg = bench.model_config((64, 64, 32), 10, 8, 0.1)
torch.manual_seed(0)
model = init_model(6, 1, "gaot_3d", cfg).to(dev).train()
opt = AdamW(model.parameters(), lr=3e-4, weight_decay=1e-5)
batch, tokens = make_synthetic_sample(500000, (64, 64, 32), k=8, seed=0, device=str(dev))
tokens = tokens.to(dev)


def step():
    gaot_3d_amd.clear_graph_cache(batch)
    opt.zero_grad(set_to_none=True)
    pred = model(batch=batch, tokens_pos=tokens)
    loss = GF.mse_loss(pred, batch.x)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host ms/step (no profiler): {(t1 - t0) / 5 * 1e3:.2f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
