#!/bin/bash
# round 4, pass u: weight-gradient GEMMs on a side stream (GAOT_DW_STREAM=1) against the single-stream step
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_u_lab.txt; : > $log
for rep in 1 2; do for sw in 0 1; do
  GAOT_DW_STREAM=$sw timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r4_u_bench_$sw.json 2> $out/r4_u_bench_$sw.err
  echo "== GAOT_DW_STREAM=$sw rc=$?" >> $log
  python3 - $out/r4_u_bench_$sw.json <<'PY' >> $log
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'launch', d['launch'], 'loss', d['loss'])
PY
  grep -i "capture failed\|error" $out/r4_u_bench_$sw.err | head -3 >> $log
done; done
GAOT_DW_STREAM=1 timeout 900 python -m pytest -q -m gpu tests/test_model_gpu.py tests/test_fullsize_gpu.py -k "not shard and not rank and not segmented" 2>&1 | grep -E "passed|failed|Error" | tail -5 >> $log
cat $log
