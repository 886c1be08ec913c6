#!/bin/bash
# round 4, pass o: fused attention backward, mask stream of a register group as ONE asm statement (LAB=132) against one statement
# per instruction group (shipped)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_o_lab.txt; : > $log
for rep in 1 2; do for lab in 0 132; do echo "== BWD LAB=$lab DROP=0.1" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=0.1 timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_bwd:" >> $log; done; done
cat $log
