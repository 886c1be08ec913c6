#!/bin/bash
# round 5, pass l: flipped-decoder list sharing (test + bench), forward row sums through v_dot2c_f32_bf16 (lab A/B)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gno_gpu.py -q -x -m gpu -k "flipped or graph" 2>&1 | tail -3
for lab in 0 8 0 8; do
  for pd in 0.1 0.0; do
    echo "== GAOT_ATTN_FWD_LAB=$lab MB_DROP=$pd"; GAOT_ATTN_FWD_LAB=$lab MB_DROP=$pd python tools/microbench.py attn 20 2>&1 | grep -E "attn_fwd|k_attn_fwd"
  done
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r5_l_bench.json 2> $out/r5_l_bench.err || tail -5 $out/r5_l_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_l_bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")}, d["geometry_cached"]["ms_per_step"], d["without_attention_dropout"]["ms_per_step"])
PY
