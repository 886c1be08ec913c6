#!/bin/bash
# round 5, pass x: the whole pass on the state after passes l-w (ABI 10: deferred reductions, shared decoder lists, dot2 row sums,
# 12-wave GNO forward): GPU suite, eager kernel stats, counter passes, the bench line; then the secondary workloads
bash tools/gpu_pass.sh r5_x
bash tools/gpu_workloads.sh r5_x cfg3 yaml cfg4 cfg4_10m cfg4_morton > $GRAFT_REPO_ROOT/gpurun_out/r5_x_workloads.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "yaml", "cfg4", "cfg4_10m", "cfg4_morton"):
    try:
        e = json.load(open(f"gpurun_out/r5_x_{w}_bench.json"))
        print(w, round(e["ms_per_step"], 2), round(e["value"] / 1e6, 1), e["roofline"]["kernel"], e["roofline"].get("traffic"))
    except Exception as ex:
        print(w, "failed", ex)
PY
