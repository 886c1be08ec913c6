#!/bin/bash
# round 5, pass ao: k_emit_sorted with four elements per trip (request-ahead): CSR / graph tests (bit-exact vs torch.sort), cfg1 / cfg4 kernel times
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -k "csr or graph or golden or cfg0 or flipped" 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -10
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $out/r5_ao_bench.json 2> $out/r5_ao_bench.err || tail -5 $out/r5_ao_bench.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r5_ao_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r5_ao_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/r5_ao_cfg4_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r5_ao_cfg4_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/r5_ao_prof $out/r5_ao_cfg4_prof -name "*_kernel_trace.csv" -delete
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $out/r5_ao_cfg4_bench.json 2> $out/r5_ao_cfg4_bench.err || tail -5 $out/r5_ao_cfg4_bench.err
python - <<'PY'
import json, csv, glob
for n in ("r5_ao_bench", "r5_ao_cfg4_bench"):
    d = json.load(open(f"gpurun_out/{n}.json")); print(n, round(d["ms_per_step"], 3), d["loss"])
for t in ("r5_ao_prof", "r5_ao_cfg4_prof"):
    f = glob.glob(f"gpurun_out/{t}/*/*kernel_stats.csv")[0]
    for r in csv.DictReader(open(f)):
        if "k_emit_sorted" in r["Name"]: print(t, r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
