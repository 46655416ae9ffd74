#!/bin/bash
# policy knobs tuned in rounds 1-3 re-checked in process on today's kernels (ViT-B)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zc; mkdir -p $O
cd $R
timeout 1500 python3 tools/ab_inproc.py gemm_groupm=0,4 gemm_groupm=0,16 gemm_groupm=0,2 gemm_groupm=0,12 gemm_splitk_xcd=1,0 attn_xcd=1,0 gemm_smallm=1,2 regions_defer=1,0 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
