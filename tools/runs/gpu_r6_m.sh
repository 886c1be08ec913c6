#!/bin/bash
# round 6, pass m: GNO backward with conflict-free 8-byte LDS stores (tile_off8) and the padded layer-0 operand: GNO tests, the SQ counter
# passes over the GNO microbenchmark (SQ_LDS_BANK_CONFLICT before: 26 M / 34 M for NH = 2 / 3, profiles/r5_an_pmc_sq.json), timing
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gno_gpu.py -q -x 2>&1 | tail -4 > $out/r6_m_tests.log; cat $out/r6_m_tests.log
bash tools/pmc_gno.sh r6_m > /dev/null 2>&1; cat $out/r6_m_pmc_sq.txt | head -60
python tools/microbench.py gno 20 > $out/r6_m_gno_microbench.txt 2>&1; tail -12 $out/r6_m_gno_microbench.txt
