#!/bin/bash
# round 3: which kernels pay when qkv / fc1 run on the four-wave GEMM kernel?  One process (tools/ab_inproc.py gemm_w4=0,1: blocks of steps A B B A ...) under rocprofv3 --kernel-trace
mkdir -p gpurun_out/r3kn
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3kn/prof -- python3 $GRAFT_REPO_ROOT/tools/ab_inproc.py gemm_w4=0,1 > $GRAFT_REPO_ROOT/gpurun_out/r3kn/ab.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3kn/ab.txt | tail -1
