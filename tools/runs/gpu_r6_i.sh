#!/bin/bash
# round 6, pass i: fused FFN forward (nothing saved) + fused backward first half (recompute) integrated: the suites that exercise it, then the bench line A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ffn_fused_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py -q -x 2>&1 | tail -8 > $out/r6_i_tests.log; cat $out/r6_i_tests.log
for f in 1 0; do
  GAOT_FFN_FUSED=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_i_bench_fused$f.json 2> $out/r6_i_bench.err || tail -5 $out/r6_i_bench.err
done
python - <<'PY'
import json
for f in (1, 0):
    e = json.load(open(f"gpurun_out/r6_i_bench_fused{f}.json"))
    print("GAOT_FFN_FUSED", f, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
