#!/bin/bash
# round 5, pass u: where the idle time of the replayed step sits (kernel trace of the hipGraph replay, gap in front of every kernel)
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/r5_u_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $out/r5_u_trace.log 2>&1
f=$(find $out/r5_u_trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/gap_sequence.py $f > $out/r5_u_gap_sequence.txt
rm -rf $out/r5_u_trace
tail -60 $out/r5_u_gap_sequence.txt
