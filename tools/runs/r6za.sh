#!/bin/bash
# side configurations: policies tuned at ViT-B re-checked in process (ViT-L with the four-wave kernel off; 6400 tokens)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6za; mkdir -p $O
cd $R
DEVIAS_GEMM_W4=0 timeout 1500 python3 tools/ab_inproc.py --model vit_large gemm_tail_split=3,4 gemm_tail_split=3,2 attn_dkdv=1,2 > $O/ab_vitl.txt 2>&1; grep -v amdgpu.ids $O/ab_vitl.txt
timeout 1500 python3 tools/ab_inproc.py --frames 32 --img-size 320 --batch 8 gemm_tail_split=3,4 attn_dkdv=1,2 gemm_w4=-1,0 > $O/ab_6400.txt 2>&1; grep -v amdgpu.ids $O/ab_6400.txt
