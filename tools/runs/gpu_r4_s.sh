#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_s_lab.txt; : > $log
timeout 600 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_boundary_gpu.py -x 2>&1 | grep -E "passed|failed|Error" | tail -5 >> $log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r4_s_bench.json 2> $out/r4_s_bench.err; echo "bench rc=$?" >> $log
python3 - <<'PY' >> $log
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_s_bench.json').read().strip().split('\n')[-1])
print('bench ms_per_step', d['ms_per_step'], 'cached', d['geometry_cached']['ms_per_step'], 'nodrop', d['without_attention_dropout']['ms_per_step'], 'roofline', d['roofline']['avg_ms'], d['roofline']['frac'])
PY
rm -rf $out/r4_s_trace
timeout 400 rocprofv3 --kernel-trace -d $out/r4_s_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/r4_s_trace.log 2>&1
python3 tools/gap_report.py $(find $out/r4_s_trace -name "*kernel_trace.csv" | head -1) 2>&1 | head -30 >> $log
cat $log
