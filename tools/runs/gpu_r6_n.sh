#!/bin/bash
# round 6, pass n: skip taps (gaot_rmsnorm_bwd2), duplicate skip inputs folded in CatLinearFn, cached lat_bidx: model / ops / sharding suites,
# the graph census of the bench (foreign kernel nodes before: 8), which ATen ops remain
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_boundary_gpu.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -4 > $out/r6_n_tests.log; cat $out/r6_n_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_n_bench.json 2> $out/r6_n_bench.err || tail -5 $out/r6_n_bench.err
python - <<'PY'
import json
e = json.load(open("gpurun_out/r6_n_bench.json"))
print(round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["foreign_kernel_nodes_per_step"], e["loss"])
PY
python tools/find_aten_kernels.py > $out/r6_n_aten.txt 2>&1; grep "aten::" $out/r6_n_aten.txt
