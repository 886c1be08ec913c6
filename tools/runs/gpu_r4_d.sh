#!/bin/bash
# round 4, pass d: stamps of the fused attention backward, GNO / attention microbenchmarks, then the full pass (tests, rocprof, PMC traffic, bench)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
log=$out/r4_d_lab.txt; : > $log
echo "== stamps (PK kernel, dropout 0.1)" >> $log
GAOT_ATTN_BWD_STAMPS=1 MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 3 2>&1 | grep -E "stamps|attn_bwd:" | tail -4 >> $log
for v in 0 7; do echo "== attn VARIANT=$v" >> $log; GAOT_ATTN_BWD_VARIANT=$v MB_DROP=0.1 MB_FUSED=1 timeout 300 python tools/microbench.py attn 10 2>&1 | grep -E "attn_bwd:|attn_fwd:" >> $log; done
echo "== gno" >> $log
timeout 300 python tools/microbench.py gno 10 2>&1 | grep -E "gno_|csr" >> $log
echo "== gemm" >> $log
timeout 300 python tools/microbench.py gemm 20 2>&1 | tail -30 >> $log
cat $log
bash tools/gpu_pass.sh r4_d
