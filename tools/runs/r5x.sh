#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; mkdir -p $O
cd $R
for i in 1 2 3; do
python3 tools/exp/fwd_time.py 2>&1 | tail -1 | tee -a $O/fwd.log
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_nofrom.so python3 tools/exp/fwd_time.py 2>&1 | tail -1 | tee -a $O/fwd.log
done
