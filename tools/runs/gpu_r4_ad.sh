#!/bin/bash
# round 4, final pass ad (after the projection-MLP backward change): all GPU tests, rocprof kernel stats + PMC traffic + bench with
# the CPU baseline, then the secondary workloads
bash tools/gpu_pass.sh r4_ad > /dev/null 2>&1
bash tools/gpu_workloads.sh r4_ad cfg3 yaml cfg4 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
tail -3 $out/r4_ad_tests.log; head -c 600 $out/r4_ad_bench.json; echo
for w in cfg3 yaml cfg4; do head -c 300 $out/r4_ad_${w}_bench.json; echo; done
