#!/bin/bash
# round 5, pass ad: the whole GPU suite and the bench line with k_attn_fwd_asm on the default path
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  GAOT_ATTN_FWD_ASM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_ad_bench_asm$v.json 2> $out/r5_ad_bench.err || tail -5 $out/r5_ad_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r5_ad_bench_asm$v.json"))
print("fwd_asm=$v", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step", "loss")}, round(d["kernels"]["attn_fwd"]["avg_ms"], 4))
PY
done
rm -f $out/r5_ad_parity.txt
GAOT_PARITY_LOG=$out/r5_ad_parity.txt python -m pytest tests -q -m gpu --maxfail=12 2>&1 | tail -25 > $out/r5_ad_tests.log
tail -8 $out/r5_ad_tests.log
