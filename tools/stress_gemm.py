import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gaot_3d_amd
from gaot_3d_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
bad = 0
for (m, n, k, c16, res) in ((16384, 256, 1024, False, True), (16384, 256, 2048, False, True), (16384, 2048, 256, True, False), (16384, 768, 256, False, False), (16384, 1024, 256, True, False), (5000, 256, 512, False, False)):
    a = torch.randn(m, k, device=dev).bfloat16(); w = torch.randn(n, k, device=dev).bfloat16()
    r = torch.randn(m, n, device=dev) if res else None
    ref = a.float() @ w.float().t() + (r if res else 0)
    tol = (2.0 ** -7 if c16 else 2e-5) * float(ref.abs().max())
    for it in range(60):
        out = ops.gemm(a, w, m, n, k, k, k, False, True, residual=r, ldr=n if res else 0, precision=1, out_dtype=torch.bfloat16 if c16 else torch.float32)
        err = float((out.float() - ref).abs().max())
        if not (err <= tol):
            bad += 1
            print("MISMATCH", m, n, k, c16, it, err, tol)
    print("shape", m, n, k, "bf16out" if c16 else "fp32out", "ok, last err", err, "tol", tol)
print("bad =", bad)
