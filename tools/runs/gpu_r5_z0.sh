#!/bin/bash
# round 5, pass z0: grid-stride row fix-up of the GNO kernels: tests, cfg1 + cfg4 bench, cfg4 kernel stats
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gno_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -10
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_z0_bench.json 2> $out/r5_z0_bench.err || tail -5 $out/r5_z0_bench.err
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r5_z0_cfg4_bench.json 2> $out/r5_z0_cfg4_bench.err || tail -5 $out/r5_z0_cfg4_bench.err
python - <<'PY'
import json
for n in ("r5_z0_bench", "r5_z0_cfg4_bench"):
    d = json.load(open(f"gpurun_out/{n}.json")); print(n, round(d["ms_per_step"], 3), d["loss"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items() if "gno" in k})
PY
