#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mhsa" > $O/t.log 2>&1; tail -5 $O/t.log
