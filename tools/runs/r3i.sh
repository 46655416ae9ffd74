#!/bin/bash
# round 3, call i: four-wave persistent GEMM: correctness + timing on the block's shapes, stream-K policy off and on
mkdir -p gpurun_out/r3i
timeout 600 python3 tools/exp/w4_check.py > gpurun_out/r3i/w4_check.txt 2>&1
SK=1 timeout 600 python3 tools/exp/w4_check.py > gpurun_out/r3i/w4_check_sk.txt 2>&1
cat gpurun_out/r3i/w4_check.txt gpurun_out/r3i/w4_check_sk.txt
