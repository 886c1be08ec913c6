#!/bin/bash
# round 4, pass i: double-buffered staging by the first half of the waves (fused attention backward)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_i_lab.txt; : > $log
for rep in 1 2 3; do for drop in 0.1 0.0; do for lab in 0 300; do echo "== BWD LAB=$lab DROP=$drop" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=$drop MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done; done; done
echo "== tests LAB=300" >> $log
GAOT_ATTN_BWD_LAB=300 timeout 900 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py tests/test_attn_dropout_gpu.py tests/test_ops_gpu.py -k "fused_backward or dropout or full_sequence or attention or attn" 2>&1 | tail -3 >> $log
cat $log
