#!/bin/bash
# round 4, pass y: GNO forward with the next tile's ids / coordinates prefetched and the f rows gathered at the top of the tile
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_y_lab.txt; : > $log
timeout 900 python -m pytest -q -m gpu tests/test_gno_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_model_gpu.py -k "gno or integral or model or configs" 2>&1 | grep -E "passed|failed|Error" | tail -5 >> $log
timeout 300 python tools/microbench.py gno 10 2>&1 | grep -E "gno_" >> $log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r4_y_bench.json 2> $out/r4_y_bench.err
timeout 600 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r4_y_bench_cfg4.json 2> $out/r4_y_bench_cfg4.err
python3 - <<'PY' >> $log
import json,os
for f in ('r4_y_bench.json','r4_y_bench_cfg4.json'):
    d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+f).read().strip().split('\n')[-1])
    print(f, 'ms_per_step', round(d['ms_per_step'],3), {k:v['avg_ms'] for k,v in d['kernels'].items() if 'gno' in k})
PY
cat $log
