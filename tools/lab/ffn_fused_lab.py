"""round 6 lab: the fused FFN forward (csrc/ffn_fused.hip) against the two launches it replaces, at the configs[1] shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops
gaot_3d_amd.set_precision("bf16")
dev = "cuda:0"
rows, f = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 1024
torch.manual_seed(0)
x = torch.randn(rows, 256, device=dev)
xb = x.bfloat16()
w13 = (torch.randn(2 * f, 256, device=dev) * 0.06)
w2 = (torch.randn(256, f, device=dev) * 0.03)
w13b, w2b = w13.bfloat16(), w2.bfloat16()


def timeit(fn, name, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / reps * 1e3
    print(f"{name}: {t:.1f} us")
    return t


t1 = timeit(lambda: ops.ffn_w13_swiglu(xb, w13b, f), "w13 + swiglu (k_gemm_k256<3>)")
ag, u = ops.ffn_w13_swiglu(xb, w13b, f)
t2 = timeit(lambda: ops.gemm(u, w2b, rows, 256, f, f, f, False, True, residual=x, ldr=256, precision=1), "w2 + residual (k_gemm_tn_n256)")
print(f"two launches: {t1 + t2:.1f} us")
packed = ops.ffn_pack(w13, w2, f, True)
timeit(lambda: ops.ffn_pack(w13, w2, f, True), "ffn_pack (fwd + bwd images)")
timeit(lambda: ops.ffn_pack(w13, w2, f, False), "ffn_pack (fwd images)")
timeit(lambda: ops.cast_bf16(w13), "cast_bf16(w13) for comparison")
timeit(lambda: ops.ffn_fwd(xb, packed, f, x), "ffn_fwd fused, a|g and u saved")
timeit(lambda: ops.ffn_fwd(xb, packed, f, x, save=False), "ffn_fwd fused, nothing saved")
# ---- backward first half ----
dy = torch.randn(rows, 256, device=dev) * 0.1
w2t = w2b.t().contiguous()
tb = timeit(lambda: ops.cast_bf16(dy), "cast_bf16(dy)")
dyb = ops.cast_bf16(dy)
tb += timeit(lambda: ops.gemm(dyb, w2t, rows, f, 256, 256, 256, False, True, precision=1, out_dtype=torch.bfloat16), "du gemm (k_gemm_k256<1>)")
du = ops.gemm(dyb, w2t, rows, f, 256, 256, 256, False, True, precision=1, out_dtype=torch.bfloat16)
tb += timeit(lambda: ops.swiglu_bwd_bf16(ag, du, f), "swiglu_bwd_bf16")
print(f"three launches: {tb:.1f} us")
timeit(lambda: ops.ffn_bwd_dag(xb, dy, packed, f), "ffn_bwd_dag fused (recompute + du + SwiGLU')")
dag = ops.swiglu_bwd_bf16(ag, du, f)
w13t = w13b.t().contiguous()
timeit(lambda: ops.gemm(dag, w13t, rows, 256, 2 * f, 2 * f, 2 * f, False, True, residual=dy, ldr=256, precision=1), "dx gemm (k_gemm_tn_n256, K = 2F)")
timeit(lambda: ops.ffn_bwd(xb, dy, packed, f, True), "ffn_bwd fused (recompute + du + SwiGLU' + dx)")
