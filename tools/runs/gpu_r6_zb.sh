#!/bin/bash
# round 6, pass zb: RMSNorm backward in the epilogues of the products in front of it (gaot_ffn_bwd_norm, gaot_qkv_bwd_norm): tests, bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ffn_fused_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py -q -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/r6_zb_tests.log; cat $out/r6_zb_tests.log
for v in 1 0 1 0; do
  GAOT_NORM_BWD_FUSED=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_zb_bench_nb$v.json 2> $out/r6_zb_bench.err || tail -5 $out/r6_zb_bench.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r6_zb_bench_nb$v.json"))
print("GAOT_NORM_BWD_FUSED", $v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
done
