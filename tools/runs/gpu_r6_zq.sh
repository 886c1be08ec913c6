#!/bin/bash
# round 6, pass zq: the fp32 one-pass attention backward's scratch cap (fallback to the two-pass kernels), attention tests
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
