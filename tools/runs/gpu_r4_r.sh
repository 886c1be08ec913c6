#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_r_gap_bisect2.txt; : > $log
run() { tag=$1; shift; rm -rf $out/r4_r_t_$tag; echo "=== $tag: $*" >> $log
  env "$@" timeout 300 rocprofv3 --kernel-trace -d $out/r4_r_t_$tag --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $EXTRA > $out/r4_r_t_$tag.log 2>&1
  python3 tools/gap_report.py $(find $out/r4_r_t_$tag -name "*kernel_trace.csv" | head -1) 2>&1 | head -16 >> $log; }
EXTRA="" run dep4096 GPU_NUM_MEM_DEPENDENCY=4096
EXTRA="" run dep16 GPU_NUM_MEM_DEPENDENCY=16
cat $log
