#!/bin/bash
# round 5, pass a: whole GPU suite on ABI 9 (global-head dropout keys, sharded == unsharded with dropout, 8 M-point GNO oracle case,
# tightened bf16 bounds) + the stream-copy forms
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
python - > $out/r5_a_stream_copy_lab.txt 2>&1 <<'PY'
import torch
from gaot_3d_amd import ops
for v in (0, 1, 2, 3, 4, 5, 6):
    r = [ops.stream_copy_gbps(variant=v) for _ in range(3)]
    print(f"stream copy variant {v}: {[round(x) for x in r]} GB/s (1 GiB, read + written)")
for v in (0, 1, 2):
    print(f"variant {v} at 256 MiB: {round(ops.stream_copy_gbps(256 << 20, 20, v))} GB/s; at 4 GiB: {round(ops.stream_copy_gbps(4 << 30, 4, v))} GB/s")
PY
rm -f $out/r5_a_parity.txt
GAOT_PARITY_LOG=$out/r5_a_parity.txt timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=15 2>&1 | tail -60 > $out/r5_a_tests.log
cat $out/r5_a_stream_copy_lab.txt; tail -40 $out/r5_a_tests.log
