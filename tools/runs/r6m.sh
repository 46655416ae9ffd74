#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O
cd $R
timeout 900 python3 tools/ab_inproc.py gemm_dynamic=0,2 gemm_dynamic=0,3 gemm_dynamic=0,2 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
for nt in 0 1 5; do echo "== fc1g, gemm_aux_nt = $nt" >> $O/pstamps.txt; DEVIAS_GEMM_AUX_NT=$nt timeout 120 python3 tools/gemm_pstamps.py fc1g >> $O/pstamps.txt 2>&1; done; grep -v amdgpu.ids $O/pstamps.txt | grep -E "==|mean|span|per work|back-to"
