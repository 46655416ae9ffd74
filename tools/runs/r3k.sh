#!/bin/bash
# round 3, call k: four-wave GEMM as the default: GEMM kernel tests, measured-path + region tests, bench A/B against the eight-wave kernels
mkdir -p gpurun_out/r3k
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r3k/tests_gemm.log 2>&1; tail -3 gpurun_out/r3k/tests_gemm.log
timeout 1500 python3 -m pytest tests/test_measured_path_gpu.py tests/test_regions_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3k/tests_path.log 2>&1; tail -3 gpurun_out/r3k/tests_path.log
for i in 1 2; do
  DEVIAS_GEMM_W4=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > gpurun_out/r3k/bench_w4_0_$i.json 2>gpurun_out/r3k/bench.err
  DEVIAS_GEMM_W4=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > gpurun_out/r3k/bench_w4_1_$i.json 2>gpurun_out/r3k/bench.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3k/bench_w4_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1]); print(f, d["value"], d["ms_per_step"])
PY
