#!/bin/bash
# One GPU-box pass: parity tests, eager rocprof kernel stats, PMC traffic passes, then the graph bench (which reports the fresh traffic).  usage: gpu_pass.sh <tag> [notests]
tag=${1:-pass}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then
  rm -f $out/${tag}_parity.txt
  GAOT_PARITY_LOG=$out/${tag}_parity.txt python -m pytest tests -q -m gpu --maxfail=12 2>&1 | tail -40 > $out/${tag}_tests.log
fi
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/${tag}_pmc_sq1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/${tag}_pmc_sq2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-secondary > $out/${tag}_pmc_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json $tag > /dev/null 2>$out/${tag}_pmc.err
python3 tools/pmc_sq.py $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 $out/${tag}_pmc_sq.json $tag > /dev/null 2>>$out/${tag}_pmc.err
cp $out/${tag}_pmc_sq.json profiles/pmc_sq.json
# the bench line reads the traffic file of THESE kernel sources (hash-stamped): refresh it before the timed run
cp $out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 10 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# keep only the small summaries of the traces
find $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write -name "*_kernel_trace.csv" -delete
find $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 -name "*_counter_collection.csv" -delete
find $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 -name "*_kernel_trace.csv" -delete
cat $out/${tag}_tests.log 2>/dev/null; head -c 1500 $out/${tag}_bench.json
