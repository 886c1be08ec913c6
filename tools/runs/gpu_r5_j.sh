#!/bin/bash
# round 5, pass j: whole GPU suite + rocprof kernel stats + PMC traffic + SQ counters + bench (with the CPU baseline) on the current tree
bash tools/gpu_pass.sh r5_j > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
tail -25 $out/r5_j_tests.log; head -c 3000 $out/r5_j_bench.json; echo; tail -5 $out/r5_j_bench.err; cat $out/r5_j_pmc.err | tail -5
