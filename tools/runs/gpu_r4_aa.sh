#!/bin/bash
# round 4, pass aa: fused attention backward, the younger wave of every SIMD offset by 64 .. 1024 cycles after the stage barrier
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_aa_lab.txt; : > $log
for drop in 0.1 0.0; do for lab in 0 203 204 205 206 207 0; do echo "== BWD LAB=$lab DROP=$drop" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=$drop timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_bwd:" >> $log; done; done
cat $log
