#!/bin/bash
# round 3, call h: GEMM time against tile rounds (this library, stream-K off / default, and the vendor library)
mkdir -p gpurun_out/r3h
DEVIAS_GEMM_SK=0 python3 tools/gemm_mscan.py > gpurun_out/r3h/mscan_dp.txt 2>&1
python3 tools/gemm_mscan.py > gpurun_out/r3h/mscan_default.txt 2>&1
python3 tools/gemm_mscan.py --vendor > gpurun_out/r3h/mscan_vendor.txt 2>&1
tail -n 8 gpurun_out/r3h/mscan_*.txt
