#!/bin/bash
# round 6, pass o: the reference's own arithmetic (fp32 end to end, pressure.yaml:4) as a measured mode: eager kernel stats of the fp32 step,
# the bench line with --precision fp32 (10 timed steps) and the fp32_mode entry of the bf16 line
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r6_o_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r6_o_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/r6_o_prof -name "*_kernel_trace.csv" -delete
cp $(find $out/r6_o_prof -name "*kernel_stats.csv" | head -1) $out/r6_o_kernel_stats_fp32.csv
head -28 $out/r6_o_kernel_stats_fp32.csv | cut -c1-120,200-330
python bench.py --precision fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/r6_o_bench_fp32.json 2> $out/r6_o_bench.err || tail -5 $out/r6_o_bench.err
python - <<'PY'
import json
e = json.load(open("gpurun_out/r6_o_bench_fp32.json"))
print(round(e["ms_per_step"], 2), e["ms_per_step_median"], e["roofline"], e["step_roofline"])
PY
