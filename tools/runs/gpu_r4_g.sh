#!/bin/bash
# round 4, pass g: 128-query stages (bf16 slots) and further schedule pins of the fused backward; schedule pins of the forward
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_g_lab.txt; : > $log
for rep in 1 2; do for drop in 0.1 0.0; do
for cfg in "0 0" "0 26" "0 100" "5 0"; do set -- $cfg; echo "== BWD VARIANT=$1 SB=$2 DROP=$drop" >> $log; GAOT_ATTN_BWD_VARIANT=$1 GAOT_ATTN_BWD_LAB=$2 MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done
for fl in 0 1 2 4 5 7; do echo "== FWD FSB=$fl DROP=$drop" >> $log; GAOT_ATTN_FWD_LAB=$fl MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_fwd:" >> $log; done
done; done
for v in 0 5; do echo "== tests VARIANT=$v" >> $log; GAOT_ATTN_BWD_VARIANT=$v timeout 900 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py tests/test_attn_dropout_gpu.py -k "fused_backward or dropout or full_sequence" 2>&1 | tail -3 >> $log; done
cat $log
