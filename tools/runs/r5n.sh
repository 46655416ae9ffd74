#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5n; mkdir -p $O
cd $R
timeout 600 python3 tools/ab_inproc.py attn_dkdv=1,2 attn_dkdv=1,3 attn_dkdv=2,3 > $O/ab.log 2>&1; tail -3 $O/ab.log
