#!/bin/bash
# round 6, pass zd: the decoder block's skip projection inside the head kernel (gaot_cat_norm_qkv_image / CatNormQKVFn): tests, the repaired
# full-graph GNO oracle cases, bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ffn_fused_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py "tests/test_fullsize_oracle_gpu.py::test_gno_full_graph_vs_oracle" -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/r6_zd_tests.log; cat $out/r6_zd_tests.log
for v in 1 0 1 0; do
  GAOT_CAT_QKV=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_zd_bench_cat$v.json 2> $out/r6_zd_bench.err || tail -5 $out/r6_zd_bench.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r6_zd_bench_cat$v.json"))
print("GAOT_CAT_QKV", $v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
done
