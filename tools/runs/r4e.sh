#!/bin/bash
mkdir -p gpurun_out/r4e
( echo "== ViT-B/16 16x224^2 B=32"; python tools/ab_inproc.py gemm_w4=0,-1 gemm_w4=0,-1
  echo "== ViT-B/16 32x320^2 (6400 tokens) B=8"; python tools/ab_inproc.py --frames 32 --img-size 320 --batch 8 gemm_w4=0,-1
  echo "== ViT-L/16 16x224^2 B=32"; python tools/ab_inproc.py --model vit_large gemm_w4=0,-1 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/ab_w4_auto.txt
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or dynamic_queue or tail_tiles" 2>&1 | tail -3
