#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O
cd $R
timeout 1200 python3 tools/ab_inproc.py gemm_ss=-1,2 gemm_ss=-1,3 > $O/ab.log 2>&1; tail -4 $O/ab.log
