#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ai; mkdir -p $O
cd $R
timeout 900 python3 tools/ab_inproc.py gemm_smallm=1,3 > $O/ab.log 2>&1; tail -1 $O/ab.log
