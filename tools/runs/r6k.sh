#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
timeout 900 python3 tools/ab_inproc.py gemm_aux_nt=5,13 gemm_aux_nt=5,21 gemm_aux_nt=5,4 > $O/ab3.txt 2>&1; grep -v amdgpu.ids $O/ab3.txt
