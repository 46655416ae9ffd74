#!/bin/bash
# round 6, call 4: weight-gradient K-tile with progressive waits (gemm_wgrad_prog), non-temporal reads of the saved pre-activation (gemm_aux_nt bit 1): tests + in-process A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
DEVIAS_GEMM_WGRAD_PROG=1 timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or persistent or gemm_layouts" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 python3 tools/ab_inproc.py gemm_wgrad_prog=0,1 gemm_aux_nt=1,3 gemm_wgrad_prog=0,1 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
