#!/bin/bash
# round 3, call o: single-launch small LayerNorm backward / column sums: kernel + region + parity tests, bench
mkdir -p gpurun_out/r3o
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm or colsum or ln_" > gpurun_out/r3o/tests_ln.log 2>&1; tail -2 gpurun_out/r3o/tests_ln.log
timeout 1500 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_recipe_gpu.py -x -q -m gpu > gpurun_out/r3o/tests_path.log 2>&1; tail -2 gpurun_out/r3o/tests_path.log
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('kernel_launches_per_step'), d.get('host_library_calls_per_step'))"; done
