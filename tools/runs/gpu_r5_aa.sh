#!/bin/bash
# round 5, pass aa: k_attn_fwd_asm (one wave per SIMD, generated tile loop): tests, microbench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "forward_asm or attention_bf16_large" 2>&1 | grep -E "^E  |passed|failed|Error|parity" | cut -c1-300 | head -20
L=$out/r5_aa_attn_fwd_asm_lab.txt; : > $L
for v in 1 0 1 0; do
  for pd in 0.1 0.0; do
    echo "== GAOT_ATTN_FWD_ASM=$v MB_DROP=$pd" >> $L; GAOT_ATTN_FWD_ASM=$v MB_DROP=$pd timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "  attn_fwd:" >> $L
  done
done
cat $L
