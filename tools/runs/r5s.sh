#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O
cd $R
timeout 900 python3 tools/ab_inproc.py attn_dkdv=2,1 > $O/ab.log 2>&1; tail -4 $O/ab.log
