#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6q; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mhsa" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 python3 tools/ab_inproc.py attn_short_last=0,1 attn_short_last=0,1 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
