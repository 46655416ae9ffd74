#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_stamp1.so timeout 300 python3 tools/exp/dkdv1w_stamps_item.py > $O/stamps.log 2>&1; cat $O/stamps.log
timeout 900 python3 tools/exp/dkdv1w_check.py time > $O/check.log 2>&1; tail -4 $O/check.log
