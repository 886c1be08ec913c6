#!/bin/bash
# round 4, pass m: fused GNO backward with four hidden layers (bf16 mode, operand fragments from L2)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_m_lab.txt; : > $log
timeout 900 python -m pytest -q -m gpu tests/test_gno_gpu.py 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 >> $log
timeout 300 python tools/microbench.py gno 5 2>&1 | grep -E "gno_|csr" >> $log
grep -E "gno_bf16_bwd_nh4|gno_shapes_bf16/c32_cd3_nh4" $out/parity_last.txt | head -20 >> $log
cat $log
