#!/bin/bash
# round 6, call 8: tail tiles as thirds / quarters (gemm_tail_split = 4): tests, in-process A/B, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python3 -m pytest tests/test_regions_gpu.py tests/test_measured_path_gpu.py -x -q > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
timeout 900 python3 tools/ab_inproc.py gemm_tail_split=2,4 gemm_tail_split=2,3 gemm_tail_split=2,4 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
for s in proj fc2 fc1g; do timeout 120 python3 tools/gemm_pstamps.py $s >> $O/pstamps.txt 2>&1; done
for nt in 0 1; do echo "== fc1g, gemm_aux_nt = $nt" >> $O/pstamps.txt; DEVIAS_GEMM_AUX_NT=$nt timeout 120 python3 tools/gemm_pstamps.py fc1g >> $O/pstamps.txt 2>&1; done; grep -v amdgpu.ids $O/pstamps.txt | grep -E "==|mean|span|per work"
