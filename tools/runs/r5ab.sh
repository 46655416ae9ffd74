#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab; mkdir -p $O
cd $R
timeout 600 python3 tools/exp/fwd1w_check.py time > $O/check.log 2>&1; grep -c "^ok" $O/check.log; tail -3 $O/check.log
