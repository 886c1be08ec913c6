#!/bin/bash
# round 6, pass zm: the S = 16 384 attention tests against the fp64 oracle with 32 / 64 / all host threads (suite time), the repaired files
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for t in 32 64 256; do
  echo "== GAOT_ORACLE_THREADS=$t" >> $out/r6_zm_attn_oracle_threads.txt
  GAOT_ORACLE_THREADS=$t timeout 900 python -m pytest tests/test_fullsize_oracle_gpu.py -q -m gpu -k "attention" --durations=10 2>&1 | grep -E "passed|failed|s call" >> $out/r6_zm_attn_oracle_threads.txt
done
cat $out/r6_zm_attn_oracle_threads.txt
timeout 900 python -m pytest tests/test_fullsize_oracle_gpu.py tests/test_ffn_fused_gpu.py -q -m gpu -k "not attention" 2>&1 | grep -E "passed|failed" 
