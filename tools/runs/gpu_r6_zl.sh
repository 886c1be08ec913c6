#!/bin/bash
# round 6, pass zl: whole GPU suite with durations on the final tree (8 M-point GNO test with its index plumbing on the device, the new
# error-behaviour test), smoke()
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
rm -f $out/r6_zl_parity.txt
GAOT_PARITY_LOG=$out/r6_zl_parity.txt timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --durations=25 2>&1 | tail -60 > $out/r6_zl_gpu_suite.txt
grep -E "passed|failed" $out/r6_zl_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/r6_zl_smoke.txt 2>&1; echo "smoke rc $?"; tail -3 $out/r6_zl_smoke.txt
