#!/bin/bash
# round 6, call 2: per-tile stamp distributions (mean / p90 / p99, per-workgroup busy time) and dgrad as NT on a transposed weight copy against the k-strided form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
timeout 300 python3 tools/exp/nt_vs_tb.py > $O/nt_vs_tb.txt 2>&1; grep -v amdgpu.ids $O/nt_vs_tb.txt
for s in qkv fc1g proj fc2 dfc2 dfc1; do timeout 120 python3 tools/gemm_pstamps.py $s >> $O/pstamps.txt 2>&1; done; grep -v amdgpu.ids $O/pstamps.txt
