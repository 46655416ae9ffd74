#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aq; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dkdv_forms" > $O/t.log 2>&1; tail -3 $O/t.log
