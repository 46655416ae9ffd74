#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 300 python3 tools/exp/wgrad_layouts.py > $O/wgrad_layouts.txt 2>&1; grep -v amdgpu.ids $O/wgrad_layouts.txt
for nt in 0 1; do echo "== fc1g, gemm_aux_nt = $nt" >> $O/pstamps.txt; DEVIAS_GEMM_AUX_NT=$nt timeout 120 python3 tools/gemm_pstamps.py fc1g >> $O/pstamps.txt 2>&1; done; grep -v amdgpu.ids $O/pstamps.txt | grep -E "==|mean|span"
