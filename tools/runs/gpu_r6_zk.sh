#!/bin/bash
# round 6, pass zk (final sources): the other BASELINE workloads on the round's sources (cfg3, yaml graph, cfg4 8 M / 10 M / Morton) with their traffic
# passes, and the N > 1 bench path end to end on ONE device over gloo (2 and 8 ranks; functional check, not a scaling number)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
bash tools/gpu_workloads.sh r6_zk cfg3 yaml cfg4 cfg4_10m cfg4_morton > $out/r6_zk_workloads.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "yaml", "cfg4", "cfg4_10m", "cfg4_morton"):
    try:
        e = json.load(open(f"gpurun_out/r6_zk_{w}_bench.json"))
        print(w, round(e["ms_per_step"], 2), round(e["value"] / 1e6, 1), e["roofline"]["kernel"], e["roofline"].get("traffic"), e["kernel_launches_per_step"])
    except Exception as ex:
        print(w, "failed", ex)
PY
for n in 2 8; do
  GAOT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline > $out/r6_zk_bench_${n}rank_one_device_gloo.json 2> $out/r6_zk_bench_${n}rank.err
  echo "rc $?"; tail -c 600 $out/r6_zk_bench_${n}rank.err | tail -3
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r6_zk_bench_${n}rank_one_device_gloo.json"))
    print($n, "ranks:", {k: d.get(k) for k in ("n_gpus", "n_ranks_seen", "ms_per_step", "loss", "launch")})
except Exception as ex:
    print($n, "ranks: no line", ex)
PY
done
