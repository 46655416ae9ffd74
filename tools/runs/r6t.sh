#!/bin/bash
# L2 prefetch, second form: the tiles that share an operand K-tile take turns (gemm_l2pf / gemm_l2pf_p > 0); < 0 = every workgroup (r6s: +0.6 / +0.5 ms)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6t; mkdir -p $O
cd $R
DEVIAS_GEMM_L2PF=2 DEVIAS_GEMM_L2PF_P=2 timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or wgrad" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 1200 python3 tools/ab_inproc.py gemm_l2pf=0,1 gemm_l2pf=0,2 gemm_l2pf=0,3 gemm_l2pf=2,0 gemm_l2pf_p=0,1 gemm_l2pf_p=0,2 gemm_l2pf_p=0,3 gemm_l2pf_p=2,0 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
