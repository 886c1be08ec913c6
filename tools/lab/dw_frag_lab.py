"""round 6 lab: the Transformer's weight-gradient products from fragment-ordered operands (csrc/dw_frag.hip) against the generic split-K
GEMM, six rotating operand sets (one set sits in the Infinity Cache and flatters every variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops
gaot_3d_amd.set_precision("bf16")
dev = "cuda:0"
rows = 16384
NSET = 6


def timeit(fn, name, reps=60):
    for i in range(6):
        fn(i % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i % NSET)
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / reps * 1e3:.1f} us")


for n1, n2, name in ((2048, 256, "dW13"), (256, 1024, "dW2"), (768, 256, "dWqkv"), (256, 256, "dWo")):
    torch.manual_seed(0)
    A = [torch.randn(rows, n1, device=dev).bfloat16() for _ in range(NSET)]
    B = [torch.randn(rows, n2, device=dev).bfloat16() for _ in range(NSET)]
    Ai = [ops.timg_pack(x) for x in A]
    Bi = [ops.timg_pack(x) for x in B]
    timeit(lambda i: ops.gemm_dw(A[i], B[i], n1, n2, rows, n1, n2, 1), f"{name} [{n1} x {n2}] generic (bf16 operands, in-call reduction)")
    timeit(lambda i: ops.dw_frag(Ai[i], Bi[i], rows, n1, n2), f"{name} [{n1} x {n2}] dw_frag (in-call reduction)")
    lib = ops._lib.load()
    splits = int(lib.gaot_dw_frag_splits(rows, n1, n2))
    part = torch.empty(splits, n1 * n2, device=dev)
    timeit(lambda i: ops.check(lib.gaot_dw_frag(ops._ptr(Ai[i]), ops._ptr(Bi[i]), rows, n1, n2, ops._ptr(part), ops._stream()), "x"),
           f"{name} [{n1} x {n2}] dw_frag kernel alone ({splits} splits)")
