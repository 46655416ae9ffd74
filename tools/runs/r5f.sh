#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
rm -f $O/abl.log
timeout 300 python3 tools/exp/dkdv1w_check.py > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log; grep -c "^ok" $O/check.log; grep "FAIL\|rc=" $O/check.log | cut -c1-200
for m in "" split1 split2 split3 "" split1 split2 split3; do
  if [ -z "$m" ]; then timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
  else DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$m.so timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1; fi
done
grep "backward with" $O/abl.log
