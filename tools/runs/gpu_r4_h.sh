#!/bin/bash
# round 4, pass h: compile-time priority variants of the fused attention backward (on the pinned schedule)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_h_lab.txt; : > $log
for rep in 1 2; do for drop in 0.1 0.0; do for lab in 0 201 202 203; do echo "== BWD LAB=$lab DROP=$drop" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done; done; done
echo "== stamps (shipped schedule)" >> $log
GAOT_ATTN_BWD_STAMPS=1 MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 2 2>&1 | grep -E "stamps" | tail -1 >> $log
cat $log
