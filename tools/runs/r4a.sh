#!/bin/bash
# round 4, first GPU call: correctness of the dynamic tile queue + CE loss + cast_scale, then A/B in process
mkdir -p gpurun_out/r4a
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or train_loss_criteria or head_match" 2>&1 | tail -15 > gpurun_out/r4a/pytest_gemm.txt
cat gpurun_out/r4a/pytest_gemm.txt
python tools/gemm_block_shapes.py DEVIAS_GEMM_DYNAMIC=0 DEVIAS_GEMM_DYNAMIC=1 DEVIAS_GEMM_DYNAMIC=0 DEVIAS_GEMM_DYNAMIC=1 > gpurun_out/r4a/block_shapes.txt 2>&1
cat gpurun_out/r4a/block_shapes.txt
python tools/ab_inproc.py gemm_dynamic=0,1 hog=16 gemm_dynamic=0,1 hog=32 gemm_dynamic=0,1 hog=0 gemm_dynamic=0,1 > gpurun_out/r4a/ab_dynamic.txt 2>&1
cat gpurun_out/r4a/ab_dynamic.txt
