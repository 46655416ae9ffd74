#!/bin/bash
# round 3, call p: final state: full GPU suite, smoke, bench lines (defaults, ViT-L, 6400 tokens), kernel stats
mkdir -p gpurun_out/r3p
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3p/smoke.log 2>&1; tail -1 gpurun_out/r3p/smoke.log
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3p/gpu_tests.log 2>&1; tail -2 gpurun_out/r3p/gpu_tests.log
python3 bench.py > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; tail -c 600 gpurun_out/r3p/bench.json
python3 bench.py --model vit_large --no-cpu-baseline --no-full-step > gpurun_out/r3p/vitl_bench.json 2>> gpurun_out/r3p/bench.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --no-cpu-baseline --no-full-step > gpurun_out/r3p/6400_bench.json 2>> gpurun_out/r3p/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3p/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3p/bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3p/prof.err
