#!/usr/bin/env python3
"""Lab: which launches of a hipGraph-replayed sequence are preceded by an idle gap?  Captures short sequences of library calls
(separated by a marker kernel), replays them under `rocprofv3 --kernel-trace`; tools/lab_gaps_report.py reads the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops, _lib
from gaot_3d_amd.ops import _ptr, _stream

dev = "cuda:0"
gaot_3d_amd.set_precision("bf16")
torch.manual_seed(0)
rows, d, f = 16384, 256, 1024
x = torch.randn(rows, d, device=dev)
w = torch.ones(d, device=dev)
xb = x.bfloat16()
w13 = torch.randn(2 * f, d, device=dev).bfloat16()
w2 = torch.randn(d, f, device=dev).bfloat16()
u = torch.randn(rows, f, device=dev).bfloat16()
state = torch.zeros(1, dtype=torch.int64, device=dev)
mark = torch.zeros(12345, device=dev)
lib = _lib.load()
ag_fix = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=dev)
u_fix = torch.empty(rows, f, dtype=torch.bfloat16, device=dev)
y_fix, r_fix = torch.empty_like(x), torch.empty(rows, device=dev)
N = 20

def seq_a():
    for _ in range(N): ops.rmsnorm_fwd(x, w, 1e-6)
def seq_b():
    for _ in range(N): ops.ffn_w13_swiglu(xb, w13, f)
def seq_c():
    for _ in range(N):
        ops.rmsnorm_fwd(x, w, 1e-6)
        ops.ffn_w13_swiglu(xb, w13, f)
def seq_d():
    for _ in range(N): ops.dropout_seed_next(state, 1)
def seq_e():
    for _ in range(N): ops.gemm(u, w2, rows, d, f, f, f, False, True)
def seq_f():     # as B, outputs preallocated (no allocator call between the launches)
    for _ in range(N):
        lib.gaot_ffn_w13_swiglu(_ptr(xb), _ptr(w13), _ptr(ag_fix), _ptr(u_fix), rows, d, d, f, _stream())
def seq_g():     # as A, outputs preallocated
    for _ in range(N):
        lib.gaot_rmsnorm_fwd(_ptr(x), _ptr(w), _ptr(y_fix), _ptr(r_fix), None, rows, d, 1e-6, _stream())
def seq_h():     # as C, outputs preallocated
    for _ in range(N):
        lib.gaot_rmsnorm_fwd(_ptr(x), _ptr(w), _ptr(y_fix), _ptr(r_fix), None, rows, d, 1e-6, _stream())
        lib.gaot_ffn_w13_swiglu(_ptr(xb), _ptr(w13), _ptr(ag_fix), _ptr(u_fix), rows, d, d, f, _stream())
def seq_i():     # torch's own kernels
    t = x
    for _ in range(N): t = t * 1.0001

def seq_j():     # true dependency, as in the model: the RMSNorm's bf16 image is the GEMM's A operand
    for _ in range(N):
        y, r, yb = ops.rmsnorm_fwd(x, w, 1e-6, want_bf16=True)
        ops.ffn_w13_swiglu(yb, w13, f)
def seq_k():     # w13 -> w2 (u produced by the first is the second's operand) -> rmsnorm of the result
    for _ in range(N):
        ag, uu = ops.ffn_w13_swiglu(xb, w13, f)
        y2 = ops.gemm(uu, w2, rows, d, f, f, f, False, True)
        ops.rmsnorm_fwd(y2, w, 1e-6, want_bf16=True)
def seq_l():     # the same chain through preallocated buffers in ONE allocation
    big = torch.empty(rows * (2 * f + f + d + d) * 4, dtype=torch.uint8, device=dev)
    o = 0
    def take(n, dt):
        nonlocal o
        t = big[o:o + n * dt.itemsize].view(dt); o += (n * dt.itemsize + 255) // 256 * 256
        return t
    agb = take(rows * 2 * f, torch.bfloat16).view(rows, 2 * f); ub = take(rows * f, torch.bfloat16).view(rows, f)
    y2 = take(rows * d, torch.float32).view(rows, d); y3 = take(rows * d, torch.float32).view(rows, d); rs = take(rows, torch.float32)
    for _ in range(N):
        lib.gaot_ffn_w13_swiglu(_ptr(xb), _ptr(w13), _ptr(agb), _ptr(ub), rows, d, d, f, _stream())
        ops.gemm(ub, w2, rows, d, f, f, f, False, True, out=y2, ldc=d)
        lib.gaot_rmsnorm_fwd(_ptr(y2), _ptr(w), _ptr(y3), _ptr(rs), None, rows, d, 1e-6, _stream())

seqs = [seq_a, seq_b, seq_c, seq_d, seq_e, seq_f, seq_g, seq_h, seq_i, seq_j, seq_k, seq_l]
def whole():
    for s in seqs:
        mark.cos_()
        s()
    mark.cos_()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    whole()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    whole()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
# the same sequence eagerly, for comparison (host-bound gaps expected)
mark.sin_()
whole()
torch.cuda.synchronize()
print("done")
