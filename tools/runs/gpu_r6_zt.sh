#!/bin/bash
# round 6, pass zt: smoke() and the default bench invocation on the final tree (what the driver runs at round end)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $out/r6_zt_smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/r6_zt_smoke.txt
t0=$(date +%s); python bench.py > $out/r6_zt_bench_default.json 2> $out/r6_zt_bench_default.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
e = json.load(open("gpurun_out/r6_zt_bench_default.json"))
print(round(e["ms_per_step"], 3), e["steps"], e["warmup"], e["roofline"]["frac"], e["roofline"]["traffic"], e["cpu_baseline"]["value"], e["fp32_mode"]["ms_per_step"], e["launch"])
PY
