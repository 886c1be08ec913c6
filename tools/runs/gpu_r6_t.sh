out=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/r6_t_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/r6_t_prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $out/r6_t_prof -name "*_kernel_trace.csv" -delete
cp $(find $out/r6_t_prof -name "*kernel_stats.csv" | head -1) $out/r6_t_kernel_stats.csv
grep "k_ffn\|k_gemm_bf16<false, true" $out/r6_t_kernel_stats.csv | cut -c1-70,200-400
