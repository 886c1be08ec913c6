#!/bin/bash
# round 4, pass c: fused attention backward after the staging-wait fixes; packed row words; key-block-outer order
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_c_attn_lab.txt; : > $log
run() { echo "== VARIANT=$1 LAB=$2 DROP=$3" >> $log; GAOT_ATTN_BWD_VARIANT=$1 GAOT_ATTN_BWD_LAB=$2 MB_DROP=$3 MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd|dq_reduce|Error|error" >> $log; }
for rep in 1 2; do
run 0 0 0.1; run 0 3 0.1; run 3 0 0.1; run 3 3 0.1; run 4 0 0.1; run 4 3 0.1; run 4 2 0.1
run 0 0 0.0; run 0 3 0.0
done
for v in 0 3 4; do
  echo "== tests VARIANT=$v" >> $log
  GAOT_ATTN_BWD_VARIANT=$v timeout 900 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py tests/test_attn_dropout_gpu.py -k "fused_backward or dropout" 2>&1 | tail -4 >> $log
done
for v in 0 4; do
GAOT_ATTN_BWD_VARIANT=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r4_c_bench_v$v.json 2> $out/r4_c_bench_v$v.err
python - <<PY >> $log
import json; o=json.load(open("$out/r4_c_bench_v$v.json")); print("bench VARIANT=$v ms/step", o["ms_per_step"], o["roofline"])
PY
done
cat $log
