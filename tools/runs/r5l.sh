#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
rm -f $O/abl.log; DEVIAS_ATTN_DQ=0 timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
for m in abl512 abl1024 abl1536 abl16 abl8 abl24; do
  DEVIAS_ATTN_DQ=0 DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$m.so timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
done
grep "backward with" $O/abl.log
