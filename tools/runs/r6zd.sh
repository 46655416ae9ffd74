#!/bin/bash
# side configurations: rasterisation group height and the weight-gradient list order re-checked in process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zd; mkdir -p $O
cd $R
timeout 1500 python3 tools/ab_inproc.py --model vit_large gemm_groupm=0,4 gemm_groupm=0,16 gemm_groupm=0,8 gemm_splitk_xcd=1,0 gemm_aux_nt=5,1 > $O/ab_vitl.txt 2>&1; grep -v amdgpu.ids $O/ab_vitl.txt
timeout 1500 python3 tools/ab_inproc.py --frames 32 --img-size 320 --batch 8 gemm_groupm=0,4 gemm_groupm=0,16 attn_xcd=1,0 > $O/ab_6400.txt 2>&1; grep -v amdgpu.ids $O/ab_6400.txt
