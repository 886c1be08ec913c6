#!/bin/bash
# Secondary bench lines (BASELINE configs[3], configs[4], the reference yaml's graph) with their PMC traffic passes.
# usage: gpu_workloads.sh <tag> [workloads...]
tag=${1:-wl}; shift
wls=${@:-cfg3 yaml cfg4}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for w in $wls; do
  # cfg4_10m = configs[4] at 10 M points; cfg4_morton = configs[4] (8 M) with the points stored along the Z-order curve
  case $w in
    cfg4_10m) wa="--workload cfg4 --points 10000000";;
    cfg4_morton) wa="--workload cfg4 --point-order morton";;
    cfg1_morton) wa="--workload cfg1 --point-order morton";;
    *) wa="--workload $w";;
  esac
  cd /tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_${w}_pmc_fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $wa --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_${w}_pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_${w}_pmc_write --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $wa --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_${w}_pmc_write.log 2>&1
  rocprofv3 --kernel-trace --stats -d $out/${tag}_${w}_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $wa --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_${w}_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/pmc_traffic.py $out/${tag}_${w}_pmc_fetch $out/${tag}_${w}_pmc_write $out/${tag}_${w}_pmc_traffic.json ${tag}_${w} > /dev/null 2>$out/${tag}_${w}_pmc.err
  cp $out/${tag}_${w}_pmc_traffic.json profiles/pmc_traffic_${w}.json
  python bench.py $wa --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_${w}_bench.json 2> $out/${tag}_${w}_bench.err
  find $out/${tag}_${w}_prof $out/${tag}_${w}_pmc_fetch $out/${tag}_${w}_pmc_write -name "*_kernel_trace.csv" -delete
  find $out/${tag}_${w}_pmc_fetch $out/${tag}_${w}_pmc_write -name "*_counter_collection.csv" -delete
  head -c 600 $out/${tag}_${w}_bench.json; echo
done
