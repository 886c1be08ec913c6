#!/bin/bash
# round 5, lab w: GNO forward variants (GAOT_GNO_FWD_VARIANT: 0 = round 4, 1 = 4 waves + full pipeline, 2 = 12 waves + id/coordinate pipeline, 3 = 12 waves)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
L=$out/r5_w_gno_fwd_variants_lab.txt; : > $L
for rep in 1 2; do
for v in 0 1 2 3; do
  echo "== GAOT_GNO_FWD_VARIANT=$v (run $rep)" >> $L
  GAOT_GNO_FWD_VARIANT=$v python tools/microbench.py gno 20 2>&1 | grep -E "gno_fwd" >> $L
done
done
cat $L
for v in 2 3; do GAOT_GNO_FWD_VARIANT=$v python -m pytest tests/test_gno_gpu.py -q -m gpu 2>&1 | tail -1; done
