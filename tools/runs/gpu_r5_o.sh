#!/bin/bash
# round 5, pass o: deferred weight-gradient reductions (ABI 10): tests, then the bench line with and without (GAOT_DEFER_REDUCE=0)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deferred_gpu.py -q -x -m gpu 2>&1 | tail -15
python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -m gpu 2>&1 | tail -4
for d in 1 0 1 0; do
  GAOT_DEFER_REDUCE=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_o_bench_defer$d.json 2> $out/r5_o_bench.err || tail -5 $out/r5_o_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r5_o_bench_defer$d.json"))
print("defer=$d", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")})
PY
done
