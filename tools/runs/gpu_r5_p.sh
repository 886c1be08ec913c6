#!/bin/bash
# round 5, pass p: deferred reductions -- tests again (storage kept instead of a view), eager kernel stats with the deferral on
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deferred_gpu.py -q -x -m gpu 2>&1 | tail -5
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r5_p_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $out/r5_p_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/r5_p_prof -name "*_kernel_trace.csv" -delete
f=$(find $out/r5_p_prof -name "*kernel_stats.csv" | head -1); cp $f $out/r5_p_kernel_stats.csv; head -30 $f | cut -c1-160
