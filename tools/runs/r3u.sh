#!/bin/bash
# round 3, call u: LayerNorm backward wave counts without spills: kernel tests, fp32 parity, ViT-L / 6400-token bench lines
mkdir -p gpurun_out/r3u
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm or ln_" > gpurun_out/r3u/tests_ln.log 2>&1; tail -1 gpurun_out/r3u/tests_ln.log
timeout 2400 python3 -m pytest tests/test_parity_gpu.py tests/test_regions_gpu.py tests/test_measured_path_gpu.py -x -q -m gpu > gpurun_out/r3u/tests_path.log 2>&1; tail -1 gpurun_out/r3u/tests_path.log
python3 bench.py --model vit_large --no-cpu-baseline --no-full-step > gpurun_out/r3u/vitl_bench.json 2> gpurun_out/r3u/bench.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --no-cpu-baseline --no-full-step > gpurun_out/r3u/6400_bench.json 2>> gpurun_out/r3u/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step > gpurun_out/r3u/bench.json 2>> gpurun_out/r3u/bench.err
python3 - <<'PY'
import json
for f in ("vitl_bench", "6400_bench", "bench"):
    d = json.loads(open(f"gpurun_out/r3u/{f}.json").read().strip().split("\n")[-1]); print(f, round(d["value"], 1), round(d["ms_per_step"], 2))
PY
