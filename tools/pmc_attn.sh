#!/bin/bash
# SQ counter passes over the attention microbenchmark (tools/microbench.py attn): usage pmc_attn.sh <tag>   (env: MB_DROP, MB_FUSED)
tag=${1:-attn}
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/${tag}_pmc1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2 > $out/${tag}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/${tag}_pmc2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2 > $out/${tag}_pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/${tag}_pmc1 k_attn_bwd > $out/${tag}_pmc_sq.txt 2>&1
python3 tools/pmc_summary.py $out/${tag}_pmc2 k_attn_bwd >> $out/${tag}_pmc_sq.txt 2>&1
find $out/${tag}_pmc1 $out/${tag}_pmc2 -name "*.csv" -delete
cat $out/${tag}_pmc_sq.txt
