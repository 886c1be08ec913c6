#!/bin/bash
# round 4, final pass x: rocprof kernel stats + PMC traffic + bench on the final sources (tests ran in pass v on the same kernels:
# only comments changed since), then the secondary workloads and the N > 1 runs on one device
bash tools/gpu_pass.sh r4_x notests > /dev/null 2>&1
bash tools/gpu_workloads.sh r4_x cfg3 yaml cfg4 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
  GAOT_BENCH_ONE_DEVICE=1 timeout 1200 python bench.py --gpus $n --steps 3 --warmup 1 --no-secondary > $out/r4_x_bench_${n}rank_one_device_gloo.json 2> $out/r4_x_bench_${n}rank.err
  tail -c 600 $out/r4_x_bench_${n}rank_one_device_gloo.json; echo
done
head -c 700 $out/r4_x_bench.json; echo
for w in cfg3 yaml cfg4; do head -c 400 $out/r4_x_${w}_bench.json; echo; done
