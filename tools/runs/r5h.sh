#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_stamp.so timeout 120 python3 tools/exp/dkdv1w_stamps.py > $O/stamps.log 2>&1; tail -3 $O/stamps.log
timeout 900 python3 tools/ab_inproc.py gemm_w4=-1,15 gemm_w4=-1,3 gemm_w4=-1,5 > $O/ab_w4.log 2>&1; tail -3 $O/ab_w4.log
