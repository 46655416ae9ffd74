#!/bin/bash
# round 3, call s: deferred, stacked weight gradients in the agg block's backward: region / parity / recipe / measured-path tests, bench
mkdir -p gpurun_out/r3s
timeout 2400 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_recipe_gpu.py tests/test_measured_path_gpu.py -x -q -m gpu > gpurun_out/r3s/tests_path.log 2>&1; tail -5 gpurun_out/r3s/tests_path.log
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; done
