#!/bin/bash
# round 4, pass w: the keep-bit image, measured as its two halves -- what WRITING the bit words costs the forward
# (GAOT_ATTN_FWD_LAB=16) and the most READING them could save the backward (GAOT_ATTN_BWD_LAB=164: masks for free)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_w_lab.txt; : > $log
for rep in 1 2; do
  for fl in 0 16; do echo "== FWD LAB=$fl DROP=0.1" >> $log; GAOT_ATTN_FWD_LAB=$fl MB_DROP=0.1 timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $log; done
  for bl in 0 164; do echo "== BWD LAB=$bl DROP=0.1" >> $log; GAOT_ATTN_BWD_LAB=$bl MB_DROP=0.1 timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_bwd:" >> $log; done
done
cat $log
