#!/bin/bash
# round 6, pass zr: fp32 attention forward with two query tiles per wave: equality against the one-tile form, tests, fp32 bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for nq in 1 2; do
GAOT_ATTN_F32_FWD_NQ=$nq python - <<PY
import torch, math, sys
sys.path.insert(0, ".")
from gaot_3d_amd import ops
outs = []
for (b, s, h, hkv, p) in ((1, 16384, 8, 8, 0.1), (2, 777, 8, 4, 0.0), (1, 333, 2, 1, 0.1), (1, 1, 1, 1, 0.0)):
    torch.manual_seed(s)
    qkv = torch.randn(b * s, (h + 2 * hkv) * 32, device="cuda:0")
    seed = torch.tensor([0x777 + s], dtype=torch.int64, device="cuda:0") if p > 0 else None
    o, lse = ops.attn_fwd(qkv, b, s, h, hkv, 1.0 / math.sqrt(32), p, seed)
    outs.append((o.cpu(), lse.cpu()))
torch.save(outs, "gpurun_out/r6_zr_fwd_nq$nq.pt")
PY
done
python - <<'PY'
import torch
a, b = torch.load("gpurun_out/r6_zr_fwd_nq1.pt"), torch.load("gpurun_out/r6_zr_fwd_nq2.pt")
print("two tiles per wave == one tile per wave, bit for bit:", all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b)))
PY
rm -f gpurun_out/r6_zr_fwd_nq1.pt gpurun_out/r6_zr_fwd_nq2.pt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attn_dropout_gpu.py tests/test_fullsize_oracle_gpu.py -q -m gpu -k "attention or attn" 2>&1 | grep -E "passed|failed" | tail -3
for v in 2 1; do
  GAOT_ATTN_F32_FWD_NQ=$v timeout 900 python bench.py --precision fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/r6_zr_bench_fp32_nq$v.json 2> $out/r6_zr_bench.err || tail -5 $out/r6_zr_bench.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r6_zr_bench_fp32_nq$v.json"))
print("GAOT_ATTN_F32_FWD_NQ", $v, round(e["ms_per_step"], 2), e["loss"])
PY
done
