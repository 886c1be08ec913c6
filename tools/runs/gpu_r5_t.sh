#!/bin/bash
# round 5, pass t: A/B of the FFN backward fusion on one box, then the whole GPU suite on ABI 10
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  GAOT_FFN_BWD_TWO_LAUNCHES=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_t_bench_two$v.json 2> $out/r5_t_bench.err || tail -5 $out/r5_t_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r5_t_bench_two$v.json"))
print("two_launches=$v", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")})
PY
done
rm -f $out/r5_t_parity.txt
GAOT_PARITY_LOG=$out/r5_t_parity.txt python -m pytest tests -q -m gpu --maxfail=12 2>&1 | tail -25 > $out/r5_t_tests.log
tail -12 $out/r5_t_tests.log
