#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ap; mkdir -p $O
cd $R
timeout 900 python3 tools/ab_inproc.py gemm_reduce_side=0,1 > $O/ab.log 2>&1; tail -2 $O/ab.log
DEVIAS_GEMM_REDUCE_SIDE=1 timeout 900 python3 -m pytest tests/test_regions_gpu.py tests/test_measured_path_gpu.py -m gpu -x -q > $O/t.log 2>&1; tail -2 $O/t.log
