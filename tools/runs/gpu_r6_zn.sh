#!/bin/bash
# round 6, pass zn: whole GPU suite on the final tree (oracle threads capped at 32 by conftest), durations
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
rm -f $out/r6_zn_parity.txt
GAOT_PARITY_LOG=$out/r6_zn_parity.txt timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --durations=25 2>&1 | tail -60 > $out/r6_zn_gpu_suite.txt
grep -E "passed|failed" $out/r6_zn_gpu_suite.txt
