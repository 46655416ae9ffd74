#!/bin/bash
# round 3, final call: last code state (tail split, stream-K policy off, small-M GEMM kernel): full GPU suite, smoke, bench lines, kernel stats
mkdir -p gpurun_out/r3fin
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3fin/smoke.log 2>&1; tail -1 gpurun_out/r3fin/smoke.log
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3fin/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r3fin/gpu_tests.log | tail -1
python3 bench.py > gpurun_out/r3fin/bench.json 2> gpurun_out/r3fin/bench.err
python3 bench.py --model vit_large --no-cpu-baseline --no-full-step > gpurun_out/r3fin/vitl_bench.json 2>> gpurun_out/r3fin/bench.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --no-cpu-baseline --no-full-step > gpurun_out/r3fin/6400_bench.json 2>> gpurun_out/r3fin/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3fin/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3fin/bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3fin/prof.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json
for f in ("bench", "vitl_bench", "6400_bench", "bench_profiled"):
    d = json.loads(open(f"gpurun_out/r3fin/{f}.json").read().strip().split("\n")[-1]); print(f, round(d["value"], 1), round(d["ms_per_step"], 2), round(d["roofline"]["frac"], 4), (d.get("full_step") or {}).get("value"))
PY
