#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ad; mkdir -p $O
cd $R
timeout 900 python3 tools/exp/dkdv1w_check.py time > $O/check.log 2>&1; grep -c "^ok" $O/check.log; grep "FAIL" $O/check.log | head -5; tail -4 $O/check.log
