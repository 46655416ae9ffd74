#!/bin/bash
mkdir -p gpurun_out/r4b
for s in qkv proj fc2; do for d in 0 1; do echo "--- $s dynamic=$d"; python tools/gemm_pstamps.py $s $d 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r4b/pstamps.txt 2>&1
grep -E "^---|dynamic queue|entry ->|epilogue \(" gpurun_out/r4b/pstamps.txt | cut -c1-200
python tools/exp/dyn_stress.py 2>&1 | grep -c "0 mismatching"
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or dynamic_queue" 2>&1 | tail -2
python tools/ab_inproc.py gemm_dynamic=0,1 hog=16 gemm_dynamic=0,1 hog=0 gemm_dynamic=0,1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b/ab_dynamic.txt
