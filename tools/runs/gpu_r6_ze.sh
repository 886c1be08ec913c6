#!/bin/bash
# round 6, pass ze: the skip projection's two input gradients inside the head's backward kernel (gaot_qkv_bwd_norm_cat): tests, bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ffn_fused_gpu.py tests/test_model_gpu.py -q 2>&1 | grep -E "passed|failed|rror|assert|parity\] cat" | tail -12 > $out/r6_ze_tests.log; cat $out/r6_ze_tests.log
for v in 1 0 1 0; do
  GAOT_CAT_BWD_DX=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_ze_bench_dx$v.json 2> $out/r6_ze_bench.err || tail -5 $out/r6_ze_bench.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r6_ze_bench_dx$v.json"))
print("GAOT_CAT_BWD_DX", $v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
done
