#!/bin/bash
# round 5, pass ai: k_attn_fwd_asm on few-head launches (key-range parts): tests, A/B per head count
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "forward_asm or attention_bf16_large" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -12
L=$out/r5_ai_attn_fwd_asm_few_heads.txt; : > $L
for h in 1 2 4 8; do
  for v in 1 0; do
    echo "== heads $h GAOT_ATTN_FWD_ASM=$v dropout 0.1" >> $L; MB_H=$h GAOT_ATTN_FWD_ASM=$v MB_DROP=0.1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_fwd:" >> $L
  done
done
cat $L
