#!/bin/bash
# round 6, pass j: eager kernel stats of the step with the fused FFN forward (what does k_ffn_fwd cost inside the step?)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r6_j_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $out/r6_j_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/r6_j_prof -name "*_kernel_trace.csv" -delete
cp $(find $out/r6_j_prof -name "*kernel_stats.csv" | head -1) $out/r6_j_kernel_stats.csv
head -30 $out/r6_j_kernel_stats.csv | cut -c1-150
