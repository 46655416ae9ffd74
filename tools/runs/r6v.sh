#!/bin/bash
# DEVIAS_ATTN_Q_PRESCALED: attention tests (new: prescaled scores), regions bitwise, parity, measured path; then in-process A/B attn_qpre
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6v; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -s -k "prescaled" > $O/t_pre.txt 2>&1; grep -E "mhsa prescaled|passed|failed|Error|assert" $O/t_pre.txt | cut -c1-400 | head -40
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mhsa" > $O/t_mhsa.txt 2>&1; tail -3 $O/t_mhsa.txt
timeout 1500 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py -x -q > $O/t_reg.txt 2>&1; tail -5 $O/t_reg.txt
timeout 1500 python3 -m pytest tests/test_measured_path_gpu.py -x -q > $O/t_meas.txt 2>&1; tail -5 $O/t_meas.txt
timeout 600 python3 tools/ab_inproc.py attn_qpre=0,1 attn_qpre=0,1 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
