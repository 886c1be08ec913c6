#!/bin/bash
# round 6, pass zs: one Transformer block at the row counts of a sharded step: row-block kernels against per-operator launches
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python tools/lab/block_rows_lab.py > $out/r6_zs_block_rows_lab.txt 2>&1; tail -12 $out/r6_zs_block_rows_lab.txt
