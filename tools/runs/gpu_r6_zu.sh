#!/bin/bash
# round 6, pass zu: last check after the host-side changes behind pass zp (fp32 backward choice in ops.attn_bwd): model / sharding / attention tests
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_attn_dropout_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
