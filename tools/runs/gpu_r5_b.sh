#!/bin/bash
# round 5, lab b: the hand-scheduled one-wave-per-SIMD fused attention backward (GAOT_ATTN_BWD_VARIANT=2) against the shipped kernel:
# results on five shapes (ragged, GQA, batches, few heads), timing at S = 16 384, H = 8 with / without dropout, fp64-oracle tests
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
L=$out/r5_b_attn_bwd_asm_lab.txt; : > $L
GAOT_ATTN_BWD_VARIANT=0 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v0.pt >> $L 2>&1
GAOT_ATTN_BWD_VARIANT=2 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v2.pt >> $L 2>&1
echo "== variant 2 (asm) vs variant 0 (shipped)" >> $L
python tools/lab/attn_bwd_variant_check.py cmp /tmp/v0.pt /tmp/v2.pt >> $L 2>&1
for v in 0 2; do for p in 0.1 0.0; do
  echo "== variant $v dropout $p" >> $L
  GAOT_ATTN_BWD_VARIANT=$v MB_DROP=$p MB_FUSED=1 timeout 300 python tools/microbench.py attn 20 2>&1 | grep -E "attn_bwd|attn_fwd" >> $L
done; done
echo "== fp64-oracle tests with variant 2" >> $L
GAOT_ATTN_BWD_VARIANT=2 timeout 900 python -m pytest tests/test_fullsize_oracle_gpu.py -q -k "attention_fused_backward" 2>&1 | tail -5 >> $L
python - >> $out/r5_b_stream_copy_lab.txt 2>&1 <<'PY'
import torch
from gaot_3d_amd import ops
for v in (1, 4, 7, 8, 9):
    r = [ops.stream_copy_gbps(variant=v) for _ in range(3)]
    print(f"stream copy variant {v}: {[round(x) for x in r]} GB/s (1 GiB, read + written)")
PY
cat $L; cat $out/r5_b_stream_copy_lab.txt
