#!/bin/bash
# round 6, pass x: ONE-box A/B of the round: every round-6 switch off (unfused FFN, no skip taps, a seed launch per block) against the defaults,
# graph replay, 20 steps each, twice (order: off, on, off, on)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  GAOT_FFN_FUSED=0 GAOT_SKIP_TAPS=0 GAOT_SEED_BLOCK=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_x_bench_off_$rep.json 2> $out/r6_x_bench.err || tail -5 $out/r6_x_bench.err
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $out/r6_x_bench_on_$rep.json 2> $out/r6_x_bench.err || tail -5 $out/r6_x_bench.err
done
python - <<'PY'
import json
for v in ("off_1", "on_1", "off_2", "on_2"):
    e = json.load(open(f"gpurun_out/r6_x_bench_{v}.json"))
    print(v, round(e["ms_per_step"], 3), e["ms_per_step_median"], e["kernel_launches_per_step"], e["graph_nodes_per_step"], e["loss"], e["roofline"]["avg_ms"])
PY
