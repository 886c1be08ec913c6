#!/bin/bash
# round 5, lab e: the asm tile loop's issue time -- the library rebuilt with GEN_LAB=nowait (no s_waitcnt in the loop) and
# GEN_LAB=nolds (no LDS instruction at all); results invalid, times only
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
L=$out/r5_e_attn_bwd_asm_issue_lab.txt; : > $L
cp gaot_3d_amd/lib/libgaot3d_hip.so /tmp/lib_orig.so
for lab in orig nowait nolds; do
  if [ $lab = orig ]; then cp /tmp/lib_orig.so gaot_3d_amd/lib/libgaot3d_hip.so; else cp tools/lab/bin/lib_$lab.so gaot_3d_amd/lib/libgaot3d_hip.so; fi
  for p in 0.1 0.0; do
    echo "== asm kernel, build $lab, dropout $p" >> $L
    GAOT_ATTN_BWD_VARIANT=2 GAOT_ATTN_BWD_LAB=3 MB_DROP=$p MB_FUSED=1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
  done
done
cp /tmp/lib_orig.so gaot_3d_amd/lib/libgaot3d_hip.so
cat $L
