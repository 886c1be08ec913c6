#!/bin/bash
# round 4, pass z: the driver's own N > 1 launch form (torch.distributed.run) with both ranks on one device over gloo; smoke()
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
GAOT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $out/r4_z_bench_2rank_torchrun.json 2> $out/r4_z_bench_2rank_torchrun.err
echo "torchrun rc=$?"; tail -c 1500 $out/r4_z_bench_2rank_torchrun.json; echo; tail -5 $out/r4_z_bench_2rank_torchrun.err | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
