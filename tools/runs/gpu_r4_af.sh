#!/bin/bash
# round 4, pass af: the oracle's attention through F.scaled_dot_product_attention when no dropout mask is given (the reference's
# own call): the two whole-step oracle tests and the bench's CPU baseline (warm-up + median) again
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_af_lab.txt; : > $log
timeout 1200 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py --durations=6 2>&1 | grep -E "^[0-9.]+s call|passed|failed" | head -10 >> $log
timeout 900 python bench.py --steps 10 --warmup 3 > $out/r4_af_bench.json 2> $out/r4_af_bench.err; echo "bench rc=$?" >> $log
python3 - <<'PY' >> $log
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_af_bench.json').read().strip().split('\n')[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'cached', round(d['geometry_cached']['ms_per_step'],3))
print('cpu_baseline', json.dumps(d['cpu_baseline'])[:900])
PY
cat $log
