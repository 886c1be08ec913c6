"""experiment: hipGraph capture of the sharded step with a 1-rank RCCL group (can torch capture its NCCL work?)"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gaot_3d_amd
from gaot_3d_amd import functional as GF, sharding
from gaot_3d_amd.data import make_synthetic_sample
from gaot_3d_amd.model import init_model
from gaot_3d_amd.optim import AdamW
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
gaot_3d_amd.set_precision("bf16")
cfg = bench.model_config((64, 64, 32), 4, 8, 0.1)
torch.manual_seed(0)
model = init_model(6, 1, "gaot_3d", cfg).to(dev).train()
opt = AdamW(model.parameters(), lr=3e-4, weight_decay=1e-5)
batch, tokens = make_synthetic_sample(100000, (64, 64, 32), k=8, seed=0, device="cuda:0")
tokens = tokens.to(dev)
local = sharding.shard_batch(batch, 0, 1, num_latent=tokens.shape[0])
ctx = sharding.ShardedStep(model, dist.group.WORLD, 100000)
# force the collective paths even with one rank
def step():
    gaot_3d_amd.clear_graph_cache(local)
    opt.zero_grad(set_to_none=True)
    loss = ctx.forward_backward(local, tokens)
    opt.step()
    return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): l0 = step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("eager loss", float(l0), flush=True)
print("capturing", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    loss = step()
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", i, float(loss))
t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); print("ms/step graph", (time.perf_counter() - t0) / 5 * 1e3)
os._exit(0)   # destroy_process_group() blocks after the group's collectives were captured
