#!/bin/bash
# round 6, pass q: ffn_norm folded into the fused FFN forward (gaot_norm_ffn_fwd, NormFFNFn): equality with norm + FFN, suites, bench A/B,
# the trimmed 8 M-point GNO cases
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ffn_fused_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py "tests/test_fullsize_oracle_gpu.py::test_gno_8m_point_graph_vs_oracle" -q -x --durations=6 2>&1 | grep -E "passed|failed|rror|^[0-9.]+s call" | tail -10 > $out/r6_q_tests.log; cat $out/r6_q_tests.log
for v in 1 0; do
  GAOT_NORM_FFN=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_q_bench_normffn$v.json 2> $out/r6_q_bench.err || tail -5 $out/r6_q_bench.err
done
python - <<'PY'
import json
for v in (1, 0):
    e = json.load(open(f"gpurun_out/r6_q_bench_normffn{v}.json"))
    print("GAOT_NORM_FFN", v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
