#!/bin/bash
# round 5, pass f: whole GPU suite + rocprof kernel stats + PMC traffic + bench (with the CPU baseline) on the current tree
bash tools/gpu_pass.sh r5_f > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
tail -25 $out/r5_f_tests.log; head -c 2500 $out/r5_f_bench.json; echo; tail -5 $out/r5_f_bench.err
