#!/bin/bash
# round 4, pass k: MFMA cluster order (S products first) and pin combinations of the fused attention backward
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_k_lab.txt; : > $log
for rep in 1 2; do for drop in 0.1 0.0; do for lab in 0 403 427 30; do echo "== BWD LAB=$lab DROP=$drop" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done; done; done
cat $log
