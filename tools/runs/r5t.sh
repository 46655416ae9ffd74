#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/exp/dkdv1w_check.py timeonly3 > $O/log.txt 2>&1
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
grep -i "dkdv\|dq_bf16" $S | cut -c1-200
rm -rf $O/trace/*/*.db
