#!/bin/bash
# round 5, pass an: the pass on the final sources (after am: GeoEmbed moments with unconditional loads):
# GPU suite, eager kernel stats, counter passes (the hash-stamped profiles/pmc_*.json), bench with the CPU baseline, smoke(), cfg4 lines
bash tools/gpu_pass.sh r5_an
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $GRAFT_REPO_ROOT/gpurun_out/r5_an_smoke.txt 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/r5_an_smoke.txt
bash tools/gpu_workloads.sh r5_an cfg3 yaml cfg4 cfg4_10m cfg4_morton > $GRAFT_REPO_ROOT/gpurun_out/r5_an_workloads.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "yaml", "cfg4", "cfg4_10m", "cfg4_morton"):
    try:
        e = json.load(open(f"gpurun_out/r5_an_{w}_bench.json"))
        print(w, round(e["ms_per_step"], 2), round(e["value"] / 1e6, 1), e["roofline"]["kernel"], e["roofline"].get("traffic"))
    except Exception as ex:
        print(w, "failed", ex)
PY
