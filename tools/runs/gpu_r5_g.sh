#!/bin/bash
# round 5, pass g: secondary workloads (configs[3], the reference yaml's graph, configs[4] at 8 M and 10 M points, configs[4] / configs[1]
# with Morton-ordered points), the N > 1 path on one device over gloo (2 / 4 / 8 ranks, overlap_dw A/B included)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
bash tools/gpu_workloads.sh r5_g cfg3 yaml cfg4 cfg4_10m cfg4_morton cfg1_morton > $out/r5_g_workloads.log 2>&1
for n in 2 4 8; do
  GAOT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 --no-secondary > $out/r5_g_bench_${n}rank_one_device_gloo.json 2> $out/r5_g_bench_${n}rank.err
  head -c 400 $out/r5_g_bench_${n}rank_one_device_gloo.json; echo; tail -3 $out/r5_g_bench_${n}rank.err
done
for w in cfg3 yaml cfg4 cfg4_10m cfg4_morton cfg1_morton; do head -c 500 $out/r5_g_${w}_bench.json; echo; done
