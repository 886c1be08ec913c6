#!/bin/bash
# round 4, pass q: what sits in the idle gaps of the hipGraph-replayed step (kernel + memory-copy + scratch traces), with the
# scratch-free fused attention backward / q|k|v image GEMM; parity of the touched kernels; bench
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_q_lab.txt; : > $log
timeout 900 python -m pytest -q -m gpu tests/test_attn_dropout_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_ops_gpu.py -k "attn or attention or fused or dropout or qkv or k256 or image" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 >> $log
rm -rf $out/r4_q_graph_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --scratch-memory-trace -d $out/r4_q_graph_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $out/r4_q_graph_trace.log 2>&1
echo "trace rc=$?" >> $log
ls $out/r4_q_graph_trace/*/ >> $log
kt=$(find $out/r4_q_graph_trace -name "*kernel_trace.csv" | head -1); mc=$(find $out/r4_q_graph_trace -name "*memory_copy_trace.csv" | head -1)
python3 tools/gap_report.py $kt $mc >> $log 2>&1
for f in $out/r4_q_graph_trace/*/*scratch_memory_trace.csv; do echo == $f >> $log; wc -l $f >> $log; head -4 $f | cut -c1-300 >> $log; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r4_q_bench.json 2> $out/r4_q_bench.err
python3 - <<'PY' >> $log
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_q_bench.json').read().strip().split('\n')[-1])
print('bench ms_per_step', d['ms_per_step'], 'roofline', d['roofline'])
PY
cat $log
