#!/bin/bash
# round 4, pass b: fused attention backward lab -- priority modes, packed row words (SDWA), one-wave-per-SIMD variants
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_b_attn_lab.txt; : > $log
run() { echo "== VARIANT=$1 LAB=$2 DROP=$3" >> $log; GAOT_ATTN_BWD_VARIANT=$1 GAOT_ATTN_BWD_LAB=$2 MB_DROP=$3 MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd|attn_fwd|dq_reduce|Error|error" >> $log; }
for rep in 1 2; do
for lab in 0 1 2 3; do run 0 $lab 0.1; done
for lab in 0 1 2 3; do run 3 $lab 0.1; done
run 1 0 0.1; run 2 0 0.1; run 1 3 0.1
run 0 0 0.0; run 0 1 0.0; run 0 3 0.0; run 1 0 0.0; run 2 0 0.0
done
# correctness of the packed-row-word variant and of the one-wave-per-SIMD variants: the fused-backward oracle tests
for v in 3 1 2; do
  echo "== tests VARIANT=$v" >> $log
  GAOT_ATTN_BWD_VARIANT=$v timeout 900 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py -k "fused_backward" 2>&1 | tail -4 >> $log
done
cat $log
