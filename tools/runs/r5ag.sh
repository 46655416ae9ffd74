#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ag; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "small_m" > $O/t.log 2>&1; tail -12 $O/t.log
