#!/bin/bash
# round 5, lab c: where k_attn_bwd_asm spends its time -- stage code alone (lab 1), tile loop without slot reduction / hashing (lab 2),
# without barriers too (lab 3); results are invalid in the lab modes, only the times count
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
L=$out/r5_c_attn_bwd_asm_split${1:+_$1}.txt; : > $L
for lab in 0 1 2 3; do for p in 0.1 0.0; do
  echo "== variant 2 lab $lab dropout $p" >> $L
  GAOT_ATTN_BWD_VARIANT=2 GAOT_ATTN_BWD_LAB=$lab MB_DROP=$p MB_FUSED=1 timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_bwd:" >> $L
done; done
cat $L
