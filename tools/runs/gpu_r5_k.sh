#!/bin/bash
# round 5, pass k: the default bench invocation as the driver runs it (CPU baseline included, wall time of the whole command), then the
# secondary workloads on the final kernels (their hash-stamped traffic files) and smoke()
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
t0=$(date +%s); python bench.py > $out/r5_k_bench.json 2> $out/r5_k_bench.err; t1=$(date +%s)
echo "default bench.py wall time: $((t1 - t0)) s" > $out/r5_k_bench_wall.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $out/r5_k_bench_wall.txt 2>&1
bash tools/gpu_workloads.sh r5_k cfg3 yaml cfg4 cfg4_10m cfg4_morton > $out/r5_k_workloads.log 2>&1
cat $out/r5_k_bench_wall.txt; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_k_bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "steps", "warmup")}, d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"]["seconds_per_step"], d["cpu_baseline"].get("bounded_sample", {}).get("seconds_by_threads"))
for w in ("cfg3", "yaml", "cfg4", "cfg4_10m", "cfg4_morton"):
    e = json.load(open(f"gpurun_out/r5_k_{w}_bench.json"))
    print(w, round(e["ms_per_step"], 2), round(e["value"] / 1e6, 1), e["roofline"]["kernel"], e["roofline"].get("traffic"))
PY
