#!/bin/bash
# round 5, pass y: GeoEmbed moments with four edges per trip, column sums with four rows in flight (both: same summation order): tests, cfg1 + cfg4
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -k "geo or colsum or linear or golden or cfg0 or deferred" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -10
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_y_bench.json 2> $out/r5_y_bench.err || tail -5 $out/r5_y_bench.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r5_y_cfg4_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r5_y_cfg4_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/r5_y_cfg4_prof -name "*_kernel_trace.csv" -delete
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r5_y_cfg4_bench.json 2> $out/r5_y_cfg4_bench.err || tail -5 $out/r5_y_cfg4_bench.err
python - <<'PY'
import json, csv, glob
for n in ("r5_y_bench", "r5_y_cfg4_bench"):
    d = json.load(open(f"gpurun_out/{n}.json")); print(n, round(d["ms_per_step"], 3), d["loss"])
f = glob.glob("gpurun_out/r5_y_cfg4_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("geo_moments", "colsum_part4", "k_gemm<4, 1, 1, 1, true")): print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
