#!/bin/bash
# LayerNorm forward: rows in flight ahead per wave (DEVIAS_LNF_PF = 1 default, 2, 3; variant builds), timed alone, processes interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zb; mkdir -p $O
cd $R
for rep in 1 2; do for v in "" lnf2 lnf3; do echo "== PF variant '${v:-default(1)}'" >> $O/lnf.txt; if [ -z "$v" ]; then timeout 120 python3 tools/exp/lnf_time.py 2>&1 | grep ln_fwd >> $O/lnf.txt; else DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$v.so timeout 120 python3 tools/exp/lnf_time.py 2>&1 | grep ln_fwd >> $O/lnf.txt; fi; done; done; cat $O/lnf.txt
