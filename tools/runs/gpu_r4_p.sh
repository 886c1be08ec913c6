#!/bin/bash
# round 4, pass p: streamed-operand weight-gradient GEMM (gemm_dw.hip) against the register-staged generic kernel, operands from HBM
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_p_lab.txt; : > $log
for sets in 1 6; do for sw in 0 1 5 2 3; do
  echo "== MB_SETS=$sets GAOT_GEMM_DW16=$sw (0 generic, 1 NB=3, 2 no compute, 3 no DMA, 4 NB=2, 5 NB=4)" >> $log
  MB_SETS=$sets GAOT_GEMM_DW16=$sw timeout 200 python3 tools/microbench.py wgrad 60 2>&1 | grep dW >> $log
done; done
cat $log
