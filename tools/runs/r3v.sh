#!/bin/bash
# round 3, call v: tail split of the persistent GEMM: bitwise check + timing alone, kernel tests, bench A/B in the step
mkdir -p gpurun_out/r3v
timeout 600 python3 tools/exp/tail_check.py > gpurun_out/r3v/tail_check.txt 2>&1
SK=1 timeout 600 python3 tools/exp/tail_check.py > gpurun_out/r3v/tail_check_sk.txt 2>&1
cat gpurun_out/r3v/tail_check.txt gpurun_out/r3v/tail_check_sk.txt | grep -v amdgpu
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r3v/tests_gemm.log 2>&1; tail -1 gpurun_out/r3v/tests_gemm.log
for i in 1 2 3; do for v in 0 1; do DEVIAS_GEMM_TAIL_SPLIT=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(\"tail_split=$v\", round(d[\"value\"],1), round(d[\"ms_per_step\"],3))"; done; done
