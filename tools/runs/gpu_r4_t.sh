#!/bin/bash
# round 4, pass t: attention forward with K / V stages by LDS-DMA (GAOT_ATTN_FWD_LAB=8) against register staging
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_t_lab.txt; : > $log
GAOT_ATTN_FWD_LAB=8 timeout 900 python -m pytest -q -m gpu tests/test_attn_dropout_gpu.py tests/test_ops_gpu.py tests/test_fullsize_oracle_gpu.py -k "attn or attention or dropout" 2>&1 | grep -E "passed|failed|Error" | tail -5 >> $log
for rep in 1 2; do for lab in 0 8; do for drop in 0.1 0.0; do echo "== FWD LAB=$lab DROP=$drop" >> $log; GAOT_ATTN_FWD_LAB=$lab MB_DROP=$drop timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $log; done; done; done
cat $log
