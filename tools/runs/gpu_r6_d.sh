#!/bin/bash
# round 6, pass d: phase stamps (s_memtime) of the fused FFN forward kernel, tests, lab against the two launches
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for b in tools/lab/bin/ffn_fwd_lab_a*; do timeout 60 $b; done > $out/r6_d_ffn_fwd_stamps.txt 2>&1
cat $out/r6_d_ffn_fwd_stamps.txt
timeout 300 python -m pytest tests/test_ffn_fused_gpu.py -q -x 2>&1 | tail -5 > $out/r6_d_tests.log; cat $out/r6_d_tests.log
timeout 120 python tools/lab/ffn_fused_lab.py > $out/r6_d_ffn_lab.txt 2>&1; cat $out/r6_d_ffn_lab.txt
