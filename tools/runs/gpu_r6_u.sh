#!/bin/bash
# round 6, pass u: the attention seeds of all blocks drawn by one launch (gaot_dropout_seed_block): dropout / model suites, bench
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_attn_dropout_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py -q -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/r6_u_tests.log; cat $out/r6_u_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_u_bench.json 2> $out/r6_u_bench.err || tail -5 $out/r6_u_bench.err
python - <<'PY'
import json
e = json.load(open("gpurun_out/r6_u_bench.json"))
print(round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
