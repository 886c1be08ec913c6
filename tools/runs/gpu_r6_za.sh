#!/bin/bash
# round 6, pass za: block-level tests of all fused forms; eager kernel stats of the step (k_norm_qkv, k_oproj_bwd, k_ffn_*)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ffn_fused_gpu.py -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > $out/r6_za_tests.log; cat $out/r6_za_tests.log
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r6_za_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r6_za_prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $out/r6_za_prof -name "*_kernel_trace.csv" -delete
cp $(find $out/r6_za_prof -name "*kernel_stats.csv" | head -1) $out/r6_za_kernel_stats.csv
grep "k_ffn\|k_norm_qkv\|k_oproj\|k_gemm\|k_rmsnorm\|k_attn" $out/r6_za_kernel_stats.csv | cut -c1-90,150-400 | head -30
