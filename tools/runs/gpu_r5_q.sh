#!/bin/bash
# round 5, pass q: deferred reductions -- which test fails; kernel stats of the GRAPH replay with the deferral on / off
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deferred_gpu.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|MISMATCH|Error" | cut -c1-300 | head -30
export TMPDIR=/tmp; cd /tmp
for d in 1 0; do
  GAOT_DEFER_REDUCE=$d rocprofv3 --kernel-trace --stats -d $out/r5_q_prof$d --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r5_q_prof$d.log 2>&1
  find $out/r5_q_prof$d -name "*_kernel_trace.csv" -delete
  f=$(find $out/r5_q_prof$d -name "*kernel_stats.csv" | head -1); cp $f $out/r5_q_kernel_stats_defer$d.csv
  tail -1 $out/r5_q_prof$d.log | cut -c1-400
done
