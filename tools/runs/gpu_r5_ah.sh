#!/bin/bash
# round 5, lab ah: forward asm variants (measurement builds): 512-key stages; MFMA positions in the unit's vector stream
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
L=$out/r5_ah_attn_fwd_asm_variants.txt; : > $L
for rep in 1 2; do
  for lab in shipped kt16 posb posc posd; do
    if [ $lab = shipped ]; then unset GAOT_LIB; else export GAOT_LIB=$GRAFT_REPO_ROOT/tools/lab/bin/lib_fwd_$lab.so; fi
    echo "== forward asm, $lab, dropout 0.1 (run $rep)" >> $L
    MB_DROP=0.1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_fwd:" >> $L
  done
done
cat $L
