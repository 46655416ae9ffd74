#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5w; mkdir -p $O
cd $R
timeout 900 python3 tools/exp/dkdv1w_check.py time > $O/check.log 2>&1; grep -c "^ok" $O/check.log; grep "FAIL\|Error\|error" $O/check.log | head; tail -4 $O/check.log
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mhsa" > $O/t.log 2>&1; tail -3 $O/t.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/exp/dkdv1w_check.py timeonly3 > $O/log.txt 2>&1
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
grep -i "dkdv\|dq_bf16" $S | cut -c1-220
rm -rf $O/trace/*/*.db
