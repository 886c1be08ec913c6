import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops
gaot_3d_amd.set_precision("bf16")
dev = "cuda:0"
rows, f = 16384, 1024
torch.manual_seed(0)
dy = torch.randn(rows, 256, device=dev).bfloat16()
w2t = (torch.randn(f, 256, device=dev) * 0.1).bfloat16()
ag = torch.randn(rows, 2 * f, device=dev).bfloat16()
def timeit(fn, name, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / reps * 1e3:.1f} us")
timeit(lambda: ops.ffn_w2_bwd_swiglu(dy, w2t, ag, f), "fused du + swiglu bwd")
timeit(lambda: ops.gemm(dy, w2t, rows, f, 256, 256, 256, False, True, precision=1, out_dtype=torch.bfloat16), "du gemm (k256<1>)")
du = ops.gemm(dy, w2t, rows, f, 256, 256, 256, False, True, precision=1, out_dtype=torch.bfloat16)
timeit(lambda: ops.swiglu_bwd_bf16(ag, du, f), "swiglu_bwd_bf16")
