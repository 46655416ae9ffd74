#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" 2>&1 | tail -1
timeout 1200 python3 tools/ab_inproc.py ln_nt=0,1 ln_nt=0,2 ln_nt=0,4 ln_nt=0,7 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
