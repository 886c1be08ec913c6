#!/bin/bash
# round 5, pass r: deferred reductions with the wide-table path: tests, bench A/B (graph replay) on one box
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deferred_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -30
for d in 1 0 1 0; do
  GAOT_DEFER_REDUCE=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_r_bench_defer$d.json 2> $out/r5_r_bench.err || tail -5 $out/r5_r_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r5_r_bench_defer$d.json"))
print("defer=$d", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")})
PY
done
