#!/bin/bash
# round 5, pass al: the N > 1 bench path end to end on ONE device over gloo (functional: spawn, eager / segmented x overlap_dw A/B under the
# watchdog, exchange_profile) on the final sources -- 2 and 8 ranks; the milliseconds are gloo staging through the host, not performance
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for n in 2 8; do
  GAOT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline > $out/r5_al_bench_${n}rank_one_device_gloo.json 2> $out/r5_al_bench_${n}rank.err
  echo "rc $?"; tail -c 600 $out/r5_al_bench_${n}rank.err | tail -3
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5_al_bench_${n}rank_one_device_gloo.json"))
    print($n, "ranks:", {k: d.get(k) for k in ("n_gpus", "n_ranks_seen", "ms_per_step", "loss", "launch")}, list((d.get("modes") or {}).keys()) if isinstance(d.get("modes"), dict) else d.get("modes"))
except Exception as ex:
    print($n, "ranks: no line", ex)
PY
done
