#!/bin/bash
# round 4, pass ac: fused projection MLP backward with dz evaluated once (transposed through a wave-private LDS tile)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_ac_lab.txt; : > $log
timeout 900 python -m pytest -q -m gpu tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_oracle_gpu.py -k "mlp2 or model or configs or projection" 2>&1 | grep -E "passed|failed|Error" | tail -5 >> $log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r4_ac_bench.json 2> $out/r4_ac_bench.err
timeout 600 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r4_ac_bench_cfg4.json 2> $out/r4_ac_bench_cfg4.err
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $out/r4_ac_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r4_ac_prof.log 2>&1; cd $GRAFT_REPO_ROOT
grep -E "k_mlp2" $(find $out/r4_ac_prof -name "*kernel_stats.csv" | head -1) | cut -c1-60,100-200 >> $log
python3 - <<'PY' >> $log
import json,os
for f in ('r4_ac_bench.json','r4_ac_bench_cfg4.json'):
    d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+f).read().strip().split('\n')[-1])
    print(f, 'ms_per_step', round(d['ms_per_step'],3), 'loss', d['loss'])
PY
cat $log
