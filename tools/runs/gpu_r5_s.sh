#!/bin/bash
# round 5, pass s: SwiGLU backward in the epilogue of the du = dy W2 product (gaot_ffn_w2_bwd_swiglu): tests, bench
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_deferred_gpu.py tests/test_model_gpu.py -q -m gpu -k "ffn or deferred or golden or cfg0 or training or ddp" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -30
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_s_bench.json 2> $out/r5_s_bench.err || tail -5 $out/r5_s_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r5_s_bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "kernel_launches_per_step")})
for k, v in d["kernels"].items():
    if "ffn" in k or "swiglu" in k: print(k, v)
PY
