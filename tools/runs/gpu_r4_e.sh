#!/bin/bash
# round 4, pass e: asymmetric priority modes of the fused attention backward
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_e_lab.txt; : > $log
for rep in 1 2; do for lab in 0 4 5 6 2; do echo "== LAB=$lab" >> $log; GAOT_ATTN_BWD_LAB=$lab MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done; done
for lab in 4 5; do echo "== stamps LAB=$lab" >> $log; GAOT_ATTN_BWD_LAB=$lab GAOT_ATTN_BWD_STAMPS=1 MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 2 2>&1 | grep -E "stamps" | tail -1 >> $log; done
echo "== no dropout" >> $log
for lab in 0 4; do GAOT_ATTN_BWD_LAB=$lab MB_DROP=0.0 MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:" >> $log; done
timeout 600 python -m pytest -q -m gpu tests/test_boundary_gpu.py 2>&1 | tail -3 >> $log
cat $log
