#!/bin/bash
# round 6, pass zj: dw_frag lab: what the time is made of (operands warm in the cache, no cross-wave sum, no loads)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
: > $out/r6_zj_dw_frag_lab.txt
for cfg in "GAOT_LAB_NSET=1" "GAOT_DW_FRAG_ABL=1" "GAOT_DW_FRAG_ABL=2" "GAOT_DW_FRAG_ABL=3" "GAOT_LAB_NSET=1 GAOT_DW_FRAG_ABL=1"; do
  echo "== $cfg" >> $out/r6_zj_dw_frag_lab.txt
  env GAOT_DW_RD=3 $cfg timeout 300 python tools/lab/dw_frag_lab.py 2>&1 | grep "kernel alone\|generic" >> $out/r6_zj_dw_frag_lab.txt
done
cat $out/r6_zj_dw_frag_lab.txt
