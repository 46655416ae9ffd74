#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
rm -f $O/abl.log
for m in "" abl4 abl64 abl128 abl192 abl256 abl260 abl63; do
  if [ -z "$m" ]; then timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
  else DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$m.so timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1; fi
done
grep "backward with" $O/abl.log
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_stamp.so timeout 120 python3 tools/exp/dkdv1w_stamps.py > $O/stamps.log 2>&1; tail -3 $O/stamps.log
