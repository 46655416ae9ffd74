#!/bin/bash
# round 5: first run of the one-wave-per-SIMD dK / dV kernel: correctness beside the old kernel and fp32, then timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout 300 python3 tools/exp/dkdv1w_check.py time > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log; tail -40 $O/check.log
