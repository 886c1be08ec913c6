#!/bin/bash
# round 6, pass k: fused FFN backward with the dx product in the launch: parity with the unfused chain, lab timing
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ffn_fused_gpu.py -q -x 2>&1 | tail -8 > $out/r6_k_tests.log; cat $out/r6_k_tests.log
timeout 120 python tools/lab/ffn_fused_lab.py > $out/r6_k_ffn_lab.txt 2>&1; cat $out/r6_k_ffn_lab.txt
