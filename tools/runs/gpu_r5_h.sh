#!/bin/bash
# round 5, pass h: the tests of what changed since pass f (k-NN beyond 64, hidden widths, overlap_dw replay, Morton reorder, RMSNorm
# backward) and a short bench
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_gno_gpu.py tests/test_loader_gpu.py tests/test_ops_gpu.py tests/test_attn_dropout_gpu.py -q -m gpu --maxfail=8 2>&1 | tail -15 > $out/r5_h_tests.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu --maxfail=8 -k "segmented or rmsnorm or golden or sample_level" 2>&1 | tail -8 >> $out/r5_h_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r5_h_bench.json 2> $out/r5_h_bench.err
cat $out/r5_h_tests.log; head -c 1200 $out/r5_h_bench.json; echo; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_h_bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")}, d["hbm_copy_peak_measured"], d["without_attention_dropout"], d["geometry_cached"])
PY
