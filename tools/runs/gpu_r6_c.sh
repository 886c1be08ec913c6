#!/bin/bash
# round 6, pass c: fused FFN forward with the chunk epilogue between the next chunk's MFMAs: tests, ablations (tools/lab/ffn_fwd_lab.hip;
# bits: 1 no weight refill, 4 no SwiGLU arithmetic, 8 no per-step LDS fragment reads, 16 no barrier), lab against the two launches
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ffn_fused_gpu.py -q -x 2>&1 | tail -5 > $out/r6_c_tests.log; cat $out/r6_c_tests.log
for b in tools/lab/bin/ffn_fwd_lab_a*; do timeout 60 $b; done > $out/r6_c_ffn_fwd_ablations.txt 2>&1
cat $out/r6_c_ffn_fwd_ablations.txt
timeout 120 python tools/lab/ffn_fused_lab.py > $out/r6_c_ffn_lab.txt 2>&1; cat $out/r6_c_ffn_lab.txt
