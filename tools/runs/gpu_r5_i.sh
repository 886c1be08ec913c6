#!/bin/bash
# round 5, lab i: the compiled two-waves-per-SIMD fused backward with its dS tile transposed on the matrix pipe (GAOT_ATTN_BWD_LAB=200)
# against the LDS round trip (lab 0), both forced with GAOT_ATTN_BWD_VARIANT=3; few-heads shapes (query-range parts) included
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
L=$out/r5_i_attn_bwd_fused_tt_lab.txt; : > $L
GAOT_ATTN_BWD_VARIANT=3 GAOT_ATTN_BWD_LAB=0 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v0.pt >> $L 2>&1
GAOT_ATTN_BWD_VARIANT=3 GAOT_ATTN_BWD_LAB=200 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v2.pt >> $L 2>&1
python tools/lab/attn_bwd_variant_check.py cmp /tmp/v0.pt /tmp/v2.pt >> $L 2>&1
for rep in 1 2; do for lab in 0 200; do for p in 0.1 0.0; do
  echo "== compiled kernel lab $lab dropout $p" >> $L
  GAOT_ATTN_BWD_VARIANT=3 GAOT_ATTN_BWD_LAB=$lab MB_DROP=$p MB_FUSED=1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
done; done; done
for h in 1 2; do for lab in 0 200; do
  echo "== $h head(s), lab $lab, dropout 0.1" >> $L
  GAOT_ATTN_BWD_VARIANT=3 GAOT_ATTN_BWD_LAB=$lab MB_H=$h MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
done; done
cat $L
