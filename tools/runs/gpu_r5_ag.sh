#!/bin/bash
# round 5, lab ag: asm backward, the unit's vector stream breadth first (two pairs at a time, in place) against pair by pair (tools/lab/bin/lib_bwd_depth.so)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_oracle_gpu.py -q -m gpu -k "attention or attn" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -12
L=$out/r5_ag_attn_bwd_order_lab.txt; : > $L
for rep in 1 2; do
  for lab in breadth depth; do
    if [ $lab = breadth ]; then unset GAOT_LIB; else export GAOT_LIB=$GRAFT_REPO_ROOT/tools/lab/bin/lib_bwd_depth.so; fi
    for pd in 0.1 0.0; do
      echo "== asm backward, order $lab, dropout $pd (run $rep)" >> $L
      MB_DROP=$pd timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
    done
  done
done
cat $L
