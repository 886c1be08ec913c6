#!/bin/bash
# round 5, pass aj: k_attn_bwd_asm on few-head launches (query-range parts): tests, A/B per head count (GAOT_ATTN_BWD_VARIANT=3 = compiled kernel)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_oracle_gpu.py -q -m gpu -k "attention or attn" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -12
L=$out/r5_aj_attn_bwd_asm_few_heads.txt; : > $L
for h in 1 2 4; do
  for v in 0 3; do
    echo "== heads $h GAOT_ATTN_BWD_VARIANT=$v dropout 0.1" >> $L; MB_H=$h GAOT_ATTN_BWD_VARIANT=$v MB_DROP=0.1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
  done
done
cat $L
