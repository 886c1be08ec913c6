#!/bin/bash
# round 6, pass p: the whole GPU suite on the current sources with per-test durations (the suite was 498 s in round 5; target <= 420 s)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
rm -f $out/r6_p_parity.txt
GAOT_PARITY_LOG=$out/r6_p_parity.txt timeout 1500 python -m pytest tests -q -m gpu --durations=30 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -60 > $out/r6_p_tests.log
tail -45 $out/r6_p_tests.log
