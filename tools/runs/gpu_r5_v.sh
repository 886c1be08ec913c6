#!/bin/bash
# round 5, pass v: GNO forward with the id / coordinate / f-row loads pipelined over the wave's tiles: tests, A/B (GAOT_GNO_FWD_NOPIPE=1 = before)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gno_gpu.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -10
L=$out/r5_v_gno_fwd_pipe_lab.txt; : > $L
for v in 0 1 0 1; do
  echo "== GAOT_GNO_FWD_NOPIPE=$v" >> $L
  GAOT_GNO_FWD_NOPIPE=$v python tools/microbench.py gno 20 2>&1 | grep -E "gno_fwd" >> $L
done
cat $L
