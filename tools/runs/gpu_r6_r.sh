#!/bin/bash
# round 6, pass r: the pass on the sources after the fused FFN (fwd / bwd with dx / norm), skip taps, GNO bank-conflict fix:
# eager kernel stats, counter passes (hash-stamped profiles/pmc_*.json), bench with the CPU baseline, smoke()
bash tools/gpu_pass.sh r6_r notests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $GRAFT_REPO_ROOT/gpurun_out/r6_r_smoke.txt 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/r6_r_smoke.txt
