#!/bin/bash
# round 6, pass a: baseline of the round on the hardened deferral (opt-in, per-task stack): the new nested-pass / DDP tests,
# which ATen ops one eager step launches (tools/find_aten_kernels.py), the bench line with the graph-node census
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deferred_gpu.py tests/test_deferred_cpu.py -q 2>&1 | tail -15 > $out/r6_a_tests.log; cat $out/r6_a_tests.log
python tools/find_aten_kernels.py > $out/r6_a_aten.txt 2>&1; tail -60 $out/r6_a_aten.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r6_a_bench.json 2> $out/r6_a_bench.err || tail -5 $out/r6_a_bench.err
python - <<'PY'
import json
e = json.load(open("gpurun_out/r6_a_bench.json"))
print(e["ms_per_step"], e["launch"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["foreign_kernel_nodes_per_step"])
print({k: e[k] for k in ("without_attention_dropout", "fp32_mode")})
PY
