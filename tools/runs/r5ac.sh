#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ac; mkdir -p $O
cd $R
python3 tools/exp/dkdv1w_check.py timeonly 2>&1 | tail -1 | tee -a $O/t.log
for m in 2048 4096 6144; do
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_abl$m.so python3 tools/exp/dkdv1w_check.py timeonly 2>&1 | tail -1 | tee -a $O/t.log
done
python3 tools/exp/dkdv1w_check.py timeonly 2>&1 | tail -1 | tee -a $O/t.log
