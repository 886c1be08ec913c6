#!/bin/bash
# round 4, pass f: compile-time sched_barrier placements in the fused attention backward (dropout and no dropout)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_f_lab.txt; : > $log
for rep in 1 2; do for drop in 0.1 0.0; do for lab in 0 1 2 8 10 11 15; do echo "== SB=$lab DROP=$drop" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done; done; done
cat $log
