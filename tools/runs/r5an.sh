#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5an; mkdir -p $O
cd $R
DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_gdbg.so timeout 1200 python3 tools/ab_inproc.py gemm_debug=0,3072 gemm_debug=0,2048 gemm_debug=0,1024 > $O/ab.log 2>&1; tail -3 $O/ab.log
