#!/bin/bash
# round 6, pass zc: the whole GPU suite on the sources with every fused block kernel on
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
rm -f $out/r6_zc_parity.txt
GAOT_PARITY_LOG=$out/r6_zc_parity.txt timeout 1500 python -m pytest tests -q -m gpu --durations=30 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -60 > $out/r6_zc_tests.log
tail -45 $out/r6_zc_tests.log
