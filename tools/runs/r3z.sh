#!/bin/bash
# round 3, call z: small-M GEMM kernel: kernel tests, path tests, bench A/B
mkdir -p gpurun_out/r3z
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "small_m" > gpurun_out/r3z/tests_smallm.log 2>&1; tail -4 gpurun_out/r3z/tests_smallm.log
timeout 2400 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_recipe_gpu.py tests/test_measured_path_gpu.py -x -q -m gpu > gpurun_out/r3z/tests_path.log 2>&1; tail -3 gpurun_out/r3z/tests_path.log
for i in 1 2 3; do for v in 0 1; do DEVIAS_GEMM_SMALLM=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(\"smallm=$v\", round(d[\"value\"],1), round(d[\"ms_per_step\"],3))"; done; done
