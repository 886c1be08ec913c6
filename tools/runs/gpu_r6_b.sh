#!/bin/bash
# round 6, pass b: fused FFN forward (csrc/ffn_fused.hip): bit equality with the two-launch path, lab timing
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ffn_fused_gpu.py -q -x 2>&1 | tail -15 > $out/r6_b_tests.log; cat $out/r6_b_tests.log
timeout 120 python tools/lab/ffn_fused_lab.py > $out/r6_b_ffn_lab.txt 2>&1; cat $out/r6_b_ffn_lab.txt
