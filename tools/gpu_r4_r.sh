#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_r_gap_bisect.txt; : > $log
run() { tag=$1; shift; rm -rf $out/r4_r_t_$tag; echo "=== $tag: $*" >> $log
  env "$@" timeout 300 rocprofv3 --kernel-trace -d $out/r4_r_t_$tag --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --layers 2 --no-cpu-baseline --no-secondary $EXTRA > $out/r4_r_t_$tag.log 2>&1
  python3 tools/gap_report.py $(find $out/r4_r_t_$tag -name "*kernel_trace.csv" | head -1) 2>&1 | head -24 >> $log; }
EXTRA="" run base A=1
EXTRA="--atten-dropout 0" run nodrop A=1
EXTRA="" run nopktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
EXTRA="" run pktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
EXTRA="" run oneq GPU_MAX_HW_QUEUES=1
cat $log
