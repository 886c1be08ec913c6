"""round 6 lab: the Transformer's weight-gradient GEMMs (bf16 operands, split-K) with 64- and 128-row tiles and forced split counts
(GAOT_DW_BM128, GAOT_DW_SPLITS are read once per process: run one process per setting)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops
gaot_3d_amd.set_precision("bf16")
dev = "cuda:0"
rows, f = 16384, 1024
torch.manual_seed(0)
dag = torch.randn(rows, 2 * f, device=dev).bfloat16()
yb = torch.randn(rows, 256, device=dev).bfloat16()
dyb = torch.randn(rows, 256, device=dev).bfloat16()
u = torch.randn(rows, f, device=dev).bfloat16()
dqkv = torch.randn(rows, 768, device=dev)
dh = torch.randn(rows, 256, device=dev)
o = torch.randn(rows, 256, device=dev)


def timeit(fn, name, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"BM128={os.environ.get('GAOT_DW_BM128', '0')} SPLITS={os.environ.get('GAOT_DW_SPLITS', '-')} {name}: {a.elapsed_time(b) / reps * 1e3:.1f} us")


timeit(lambda: ops.gemm_dw(dag, yb, 2 * f, 256, rows, 2 * f, 256, 1), "dW13 [2048 x 256]")
timeit(lambda: ops.gemm_dw(dyb, u, 256, f, rows, 256, f, 1), "dW2 [256 x 1024]")
timeit(lambda: ops.gemm_dw(dqkv, yb, 768, 256, rows, 768, 256, 1), "dWqkv [768 x 256] (fp32 dqkv)")
timeit(lambda: ops.gemm_dw(dh, o, 256, 256, rows, 256, 256, 1), "dWo [256 x 256] (fp32 operands)")
