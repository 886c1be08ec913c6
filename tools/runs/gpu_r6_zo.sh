#!/bin/bash
# round 6, pass zo: fp32 mode, the one-pass attention backward (gaot_attn_bwd_fused_f32): parity tests, fp32-mode bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attn_dropout_gpu.py tests/test_fullsize_oracle_gpu.py -q -m gpu -k "attention or attn" 2>&1 | grep -E "passed|failed|rror|assert|one_pass" | tail -14 > $out/r6_zo_tests.log; cat $out/r6_zo_tests.log
for v in 1 0; do
  GAOT_ATTN_F32_FUSED=$v timeout 900 python bench.py --precision fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/r6_zo_bench_fp32_fused$v.json 2> $out/r6_zo_bench.err || tail -5 $out/r6_zo_bench.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r6_zo_bench_fp32_fused$v.json"))
print("GAOT_ATTN_F32_FUSED", $v, round(e["ms_per_step"], 2), e["loss"], e.get("roofline"))
PY
done
