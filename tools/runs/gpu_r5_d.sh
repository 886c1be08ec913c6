#!/bin/bash
# round 5, lab d: k_attn_bwd_asm (GAOT_ATTN_BWD_VARIANT=2) against the compiled fused backward (=3): results on five shapes, timing
# (two rounds each, same box); generator parameters vary between runs, see profiles/README.md
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
L=$out/r5_d_attn_bwd_asm_${1:-x}.txt; : > $L
GAOT_ATTN_BWD_VARIANT=3 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v0.pt >> $L 2>&1
GAOT_ATTN_BWD_VARIANT=2 timeout 300 python tools/lab/attn_bwd_variant_check.py run /tmp/v2.pt >> $L 2>&1
python tools/lab/attn_bwd_variant_check.py cmp /tmp/v0.pt /tmp/v2.pt >> $L 2>&1
for v in 3 2 3 2; do for p in 0.1 0.0; do
  echo "== variant $v dropout $p" >> $L
  GAOT_ATTN_BWD_VARIANT=$v MB_DROP=$p MB_FUSED=1 timeout 300 python tools/microbench.py attn 30 2>&1 | grep -E "  attn_bwd:" >> $L
done; done
cat $L
if [ "$2" = tests ]; then
  timeout 1200 python -m pytest tests/test_fullsize_oracle_gpu.py tests/test_attn_dropout_gpu.py tests/test_ops_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -4
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ('ms_per_step','ms_per_step_median','loss')}, d['roofline'])"
fi
