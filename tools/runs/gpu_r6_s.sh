#!/bin/bash
# round 6, pass s: the whole block tail in one forward launch (gaot_block_tail_fwd / BlockTailFn): tests, suites, bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ffn_fused_gpu.py tests/test_model_gpu.py tests/test_deferred_gpu.py -q -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/r6_s_tests.log; cat $out/r6_s_tests.log
for v in 1 0; do
  GAOT_BLOCK_TAIL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_s_bench_tail$v.json 2> $out/r6_s_bench.err || tail -5 $out/r6_s_bench.err
done
python - <<'PY'
import json
for v in (1, 0):
    e = json.load(open(f"gpurun_out/r6_s_bench_tail{v}.json"))
    print("GAOT_BLOCK_TAIL", v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"])
PY
