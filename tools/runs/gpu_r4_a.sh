#!/bin/bash
# round 4, pass a: the new variant / config tests, the bench line's new fields, the N>1 exchange profile on one device
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
GAOT_PARITY_LOG=$out/r4_a_parity.txt timeout 1500 python -m pytest -q -m gpu --maxfail=20 \
  tests/test_gno_gpu.py::test_integral_transform_variants_golden \
  "tests/test_graph_gpu.py" \
  tests/test_fullsize_oracle_gpu.py::test_configs3_full_size_vs_oracle \
  tests/test_fullsize_gpu.py::test_configs4_point_sharded_two_ranks_one_gpu \
  "tests/test_fullsize_oracle_gpu.py::test_attention_full_sequence_vs_fp64_oracle" \
  "tests/test_fullsize_oracle_gpu.py::test_attention_fused_backward_vs_fp64_oracle" \
  tests/test_edgeops_gpu.py 2>&1 | tail -60 > $out/r4_a_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r4_a_bench.json 2> $out/r4_a_bench.err
GAOT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary > $out/r4_a_bench_2rank_one_device_gloo.json 2> $out/r4_a_bench_2rank.err
cat $out/r4_a_tests.log; head -c 600 $out/r4_a_bench.json; echo; tail -5 $out/r4_a_bench.err; head -c 300 $out/r4_a_bench_2rank_one_device_gloo.json; tail -5 $out/r4_a_bench_2rank.err
