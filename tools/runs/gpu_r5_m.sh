#!/bin/bash
# round 5, pass m: forward mask polarity (v_and) + dot2 row sums as the dropout default, asm backward dS = p(-delta') + (keep p) dP' (mul + fmac);
# instruction issue costs (tools/lab/inst_cost.hip)
out=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
tools/lab/bin/inst_cost > $out/r5_m_inst_cost.txt 2>&1
python -m pytest tests -q -x -m gpu -k "attn or attention or dropout" 2>&1 | tail -4
for pd in 0.1 0.0; do
  echo "== MB_DROP=$pd"; MB_DROP=$pd python tools/microbench.py attn 20 2>&1 | grep -E "attn_fwd|attn_bwd|k_attn"
done
echo "== adds forced"; GAOT_ATTN_FWD_LAB=1 MB_DROP=0.1 python tools/microbench.py attn 20 2>&1 | grep -E "attn_fwd"
