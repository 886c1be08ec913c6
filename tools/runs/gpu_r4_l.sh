#!/bin/bash
# round 4, pass l: fused attention backward with query-range parts for few heads per launch (the per-rank shapes of a sharded step)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_l_lab.txt; : > $log
timeout 900 python -m pytest -q -m gpu tests/test_fullsize_oracle_gpu.py -k "fused_backward" 2>&1 | tail -4 >> $log
for h in 1 2 4 8; do for fused in 1 0; do for drop in 0.1; do echo "== H=$h FUSED=$fused DROP=$drop" >> $log; MB_H=$h MB_DROP=$drop MB_FUSED=$fused timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd|attn_fwd:" >> $log; done; done; done
timeout 900 python -m pytest -q -m gpu tests/test_model_gpu.py -k "shard or seq or rank or segmented" 2>&1 | tail -3 >> $log
cat $log
