#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5am; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_measured_path_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log; grep "AssertionError" $O/t.log | head -3
timeout 900 python3 tools/ab_inproc.py regions_defer=1,1 > $O/ab.log 2>&1; tail -1 $O/ab.log
