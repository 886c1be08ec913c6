#!/bin/bash
# round 6, pass l: fused FFN backward with dx in the launch, integrated: the suites that exercise it, bench A/B (GAOT_FFN_BWD_DX, GAOT_FFN_FUSED)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ffn_fused_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py -q -x 2>&1 | tail -4 > $out/r6_l_tests.log; cat $out/r6_l_tests.log
for v in "1 1" "1 0" "0 0"; do set -- $v
  GAOT_FFN_FUSED=$1 GAOT_FFN_BWD_DX=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_l_bench_f$1_dx$2.json 2> $out/r6_l_bench.err || tail -5 $out/r6_l_bench.err
done
python - <<'PY'
import json
for v in ("f1_dx1", "f1_dx0", "f0_dx0"):
    e = json.load(open(f"gpurun_out/r6_l_bench_{v}.json"))
    print(v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
