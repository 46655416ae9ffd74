#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zg; mkdir -p $O
cd $R
timeout 600 python3 tools/torch_ops_in_step.py > $O/ops.txt 2>&1; grep -v amdgpu.ids $O/ops.txt | tail -48
