#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aa; mkdir -p $O
cd $R
timeout 600 python3 tools/exp/fwd1w_check.py time > $O/check.log 2>&1; tail -20 $O/check.log
