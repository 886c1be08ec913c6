#!/bin/bash
# round 4, pass j: the whole tree -- all GPU tests, rocprof kernel stats, PMC traffic, bench (with the CPU baseline), SQ counters of
# the fused attention backward, the secondary workloads, and the N > 1 bench path on one device (2 / 4 / 8 ranks over gloo)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
bash tools/gpu_pass.sh r4_j > $out/r4_j_pass.log 2>&1
python bench.py --steps 20 --warmup 5 > $out/r4_j_bench_full.json 2> $out/r4_j_bench_full.err
MB_DROP=0.1 MB_FUSED=1 bash tools/pmc_attn.sh r4_j_attn_bwd_fused > /dev/null 2>&1
bash tools/gpu_workloads.sh r4_j cfg3 yaml cfg4 > $out/r4_j_workloads.log 2>&1
for n in 2 4 8; do
  GAOT_BENCH_ONE_DEVICE=1 timeout 1200 python bench.py --gpus $n --steps 3 --warmup 1 --no-secondary > $out/r4_j_bench_${n}rank_one_device_gloo.json 2> $out/r4_j_bench_${n}rank.err
done
tail -3 $out/r4_j_tests.log; head -c 400 $out/r4_j_bench_full.json; echo; head -c 300 $out/r4_j_cfg4_bench.json; echo; head -c 200 $out/r4_j_bench_8rank_one_device_gloo.json
