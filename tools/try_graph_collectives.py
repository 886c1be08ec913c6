"""experiment: which torch.distributed collectives of the sharded step survive hipGraph capture + replay on RCCL
(1-rank group on a one-GPU box; run with python -u)"""
import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
x = torch.randn(1024, 256, device=dev)

def ar(): y = x.clone(); dist.all_reduce(y); return y
def ag(): out = torch.empty(1, 1024, 256, device=dev); dist.all_gather_into_tensor(out, x); return out
def rs(): out = torch.empty(1024, 256, device=dev); dist.reduce_scatter_tensor(out, x); return out
def a2a(): out = torch.empty_like(x); dist.all_to_all_single(out, x); return out

for name, fn in (("all_reduce", ar), ("all_gather_into_tensor", ag), ("reduce_scatter_tensor", rs), ("all_to_all_single", a2a)):
    print("eager", name, flush=True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    print("capture", name, flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = fn()
        print("replay", name, flush=True)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print("ok", name, float(y.abs().sum()), flush=True)
    except Exception as ex:
        print("FAILED", name, type(ex).__name__, str(ex)[:300], flush=True)
        torch.cuda.synchronize()
print("done", flush=True)
os._exit(0)
