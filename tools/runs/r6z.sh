#!/bin/bash
# ViT-L (config 4): is the four-wave kernel still the better server of its GEMMs after round 6's work on the eight-wave kernel (specialised epilogues, tail thirds)?  Its N = 1024 shapes have 3.06 rounds of tiles.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
timeout 1500 python3 tools/ab_inproc.py --model vit_large gemm_w4=-1,0 gemm_w4=-1,0 gemm_w4=0,1 gemm_w4=0,2 > $O/ab_vitl.txt 2>&1; grep -v amdgpu.ids $O/ab_vitl.txt
