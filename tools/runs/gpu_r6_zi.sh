#!/bin/bash
# round 6, pass zi: dw_frag lab with device timing (graph replay): ring depth, split cap
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
: > $out/r6_zi_dw_frag_lab.txt
for rd in 3 5; do
  echo "== GAOT_DW_RD=$rd" >> $out/r6_zi_dw_frag_lab.txt
  GAOT_DW_RD=$rd timeout 300 python tools/lab/dw_frag_lab.py 2>&1 | grep "us$" >> $out/r6_zi_dw_frag_lab.txt
done
for cap in 8 16 32; do
  echo "== GAOT_DW_RD=3 GAOT_DW_FRAG_MAXSPLITS=$cap" >> $out/r6_zi_dw_frag_lab.txt
  GAOT_DW_RD=3 GAOT_DW_FRAG_MAXSPLITS=$cap timeout 300 python tools/lab/dw_frag_lab.py 2>&1 | grep "dw_frag" >> $out/r6_zi_dw_frag_lab.txt
done
cat $out/r6_zi_dw_frag_lab.txt
