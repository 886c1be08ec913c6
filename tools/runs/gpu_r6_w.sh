#!/bin/bash
# round 6, pass w: lab -- tile height and split count of the split-K weight-gradient GEMMs (bf16 operands)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for bm in 0 1; do for sp in 0 4 8 16 32; do
  if [ $sp = 0 ]; then GAOT_DW_BM128=$bm timeout 120 python tools/lab/dw_gemm_lab.py; else GAOT_DW_BM128=$bm GAOT_DW_SPLITS=$sp timeout 120 python tools/lab/dw_gemm_lab.py; fi
done; done 2>&1 | grep "BM128" > $out/r6_w_dw_gemm_lab.txt
cat $out/r6_w_dw_gemm_lab.txt
