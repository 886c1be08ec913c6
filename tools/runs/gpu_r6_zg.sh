#!/bin/bash
# round 6, pass zg: dw_frag (weight gradients from fragment-ordered operands): parity tests + lab timing against the generic split-K GEMM
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dw_frag_gpu.py -q 2>&1 | grep -E "passed|failed|rror|assert|parity\]" | tail -12 > $out/r6_zg_tests.log; cat $out/r6_zg_tests.log
timeout 600 python tools/lab/dw_frag_lab.py > $out/r6_zg_dw_frag_lab.txt 2>&1; cat $out/r6_zg_dw_frag_lab.txt | tail -14
