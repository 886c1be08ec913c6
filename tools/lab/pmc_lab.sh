#!/bin/bash
# SQ counters over one lab binary: pmc_lab.sh <tag> <binary> [args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/$1; shift
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/${tag}_pmc1 --output-format csv -- $B "$@" > $out/${tag}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/${tag}_pmc2 --output-format csv -- $B "$@" > $out/${tag}_pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/${tag}_pmc1 k_gemm > $out/${tag}_pmc_sq.txt 2>&1
python3 tools/pmc_summary.py $out/${tag}_pmc2 k_gemm >> $out/${tag}_pmc_sq.txt 2>&1
find $out/${tag}_pmc1 $out/${tag}_pmc2 -name "*.csv" -delete
cat $out/${tag}_pmc_sq.txt
