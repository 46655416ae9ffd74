#!/bin/bash
mkdir -p gpurun_out/r4d
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or dynamic_queue" 2>&1 | tail -2
python tools/exp/dyn_stress.py 2>&1 | grep -c "0 mismatching"
python tools/ab_inproc.py gemm_dynamic=0,1 hog=16 gemm_dynamic=0,1 gemm_reserve_cus=0,16 hog=0 gemm_reserve_cus=0,16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d/ab_dynamic.txt
