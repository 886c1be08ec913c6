#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
MB_DROP=0.1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $out/fwd_pmc1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2 > $out/fwd_pmc1.log 2>&1
MB_DROP=0.1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $out/fwd_pmc2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2 > $out/fwd_pmc2.log 2>&1
MB_DROP=0.1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $out/fwd_pmc3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2 > $out/fwd_pmc3.log 2>&1
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 tools/pmc_summary.py $out/fwd_pmc$i "k_attn_fwd_bf16<4" ; done > $out/attn_fwd_pmc_sq.txt 2>&1
find $out/fwd_pmc1 $out/fwd_pmc2 $out/fwd_pmc3 -name "*.csv" -delete
cat $out/attn_fwd_pmc_sq.txt
tail -3 $out/fwd_pmc3.log
