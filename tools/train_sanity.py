#!/usr/bin/env python3
"""Does the optimised step TRAIN?  configs[1] (500 000 points, L = 10, bf16 kernels, attention dropout 0.1), a learnable target
(a smooth function of position and normal), AdamW, the whole step replayed from ONE captured hipGraph as bench.py does.
Prints the loss every few steps; the loss must fall well below the variance of the target.  usage: train_sanity.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaot_3d_amd
from gaot_3d_amd import functional as GF
from gaot_3d_amd.data import make_synthetic_sample
from gaot_3d_amd.model import init_model
from gaot_3d_amd.optim import AdamW
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
gaot_3d_amd.set_precision("bf16")
cfg = bench.model_config((64, 64, 32), 10, 8, 0.1)
torch.manual_seed(0)
model = init_model(6, 1, "gaot_3d", cfg).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
batch, tokens = make_synthetic_sample(500000, (64, 64, 32), k=8, seed=0, device=str(dev))
tokens = tokens.to(dev)
p, nrm = batch.pos, batch.c if hasattr(batch, "c") else batch.pos
t = torch.sin(3.0 * p[:, :1]) * torch.cos(2.0 * p[:, 1:2]) + 0.5 * p[:, 2:3] + 0.3 * nrm[:, :1]
batch.x = ((t - t.mean()) / t.std()).contiguous()
loss_buf = torch.zeros((), device=dev)


def step():
    gaot_3d_amd.clear_graph_cache(batch)
    opt.zero_grad(set_to_none=True)
    pred = model(batch=batch, tokens_pos=tokens)
    loss = GF.mse_loss(pred, batch.x)
    loss.backward()
    opt.step()
    loss_buf.copy_(loss.detach())


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(f"eager warm-up, loss after 3 steps: {float(loss_buf):.5f}")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
hist = []
for i in range(steps):
    g.replay()
    if i % 5 == 4 or i == steps - 1:
        torch.cuda.synchronize()
        hist.append((i + 4, float(loss_buf)))
        print(f"step {i + 4:3d}  loss {hist[-1][1]:.5f}")
assert hist[-1][1] < 0.5 * hist[0][1] or hist[-1][1] < 0.2, "the loss did not fall"
print("OK: the replayed bf16 step with dropout trains (target variance 1.0)")
