#!/bin/bash
# round 5, lab ac: where the forward asm kernel's time is (results of the lab builds are invalid): stage code alone, loop without MFMAs, loop without vector instructions
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
L=$out/r5_ac_attn_fwd_asm_split.txt; : > $L
for pd in 0.1 0.0; do
  echo "== shipped, dropout $pd" >> $L; MB_DROP=$pd timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $L
  echo "== stage code alone (GAOT_ATTN_FWD_ASM_LAB=1), dropout $pd" >> $L; GAOT_ATTN_FWD_ASM_LAB=1 MB_DROP=$pd timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $L
  for lab in nomfma; do
    echo "== tile loop $lab, dropout $pd" >> $L; GAOT_LIB=$GRAFT_REPO_ROOT/tools/lab/bin/lib_fwd_$lab.so MB_DROP=$pd timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $L
  done
done
cat $L
