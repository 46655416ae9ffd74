#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6w; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -s -k "prescaled" > $O/t_pre.txt 2>&1; grep -E "mhsa prescaled|passed|failed|Error|assert" $O/t_pre.txt | cut -c1-700 | head -40
