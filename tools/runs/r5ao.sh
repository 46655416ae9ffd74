#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ao; mkdir -p $O
cd $R
for i in 1 2; do
python3 tools/exp/dkdv1w_check.py timeonly 2>&1 | tail -1 | tee -a $O/t.log
DEVIAS_DKDV_NW2=1 python3 tools/exp/dkdv1w_check.py timeonly 2>&1 | tail -1 | tee -a $O/t.log
done
DEVIAS_DKDV_NW2=1 timeout 600 python3 tools/exp/dkdv1w_check.py 2>&1 | tail -2 | tee -a $O/t.log
