#!/bin/bash
# round 5, lab n: asm backward dS form A/B on one box -- mul + fmac (shipped) against the two-select form (tools/lab/bin/lib_ds_select.so)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
L=$out/r5_n_attn_bwd_ds_form_lab.txt; : > $L
for rep in 1 2; do
  for lab in fmac select; do
    if [ $lab = fmac ]; then unset GAOT_LIB; else export GAOT_LIB=$GRAFT_REPO_ROOT/tools/lab/bin/lib_ds_select.so; fi
    echo "== asm backward, dS form $lab, dropout 0.1 (run $rep)" >> $L
    MB_DROP=0.1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:|  attn_fwd:" >> $L
  done
done
cat $L
