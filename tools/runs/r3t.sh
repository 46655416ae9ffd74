#!/bin/bash
# round 3, call t: final state (after the LayerNorm and agg-block changes): full GPU suite, smoke, bench lines (defaults, ViT-L, 6400 tokens), kernel stats
mkdir -p gpurun_out/r3t
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3t/smoke.log 2>&1; tail -1 gpurun_out/r3t/smoke.log
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3t/gpu_tests.log 2>&1; tail -2 gpurun_out/r3t/gpu_tests.log
python3 bench.py > gpurun_out/r3t/bench.json 2> gpurun_out/r3t/bench.err; tail -c 600 gpurun_out/r3t/bench.json
python3 bench.py --model vit_large --no-cpu-baseline --no-full-step > gpurun_out/r3t/vitl_bench.json 2>> gpurun_out/r3t/bench.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --no-cpu-baseline --no-full-step > gpurun_out/r3t/6400_bench.json 2>> gpurun_out/r3t/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3t/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3t/bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3t/prof.err
cd $GRAFT_REPO_ROOT
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-full-step --force-gradsync > gpurun_out/r3t/bench_torchrun_forcesync.json 2> gpurun_out/r3t/torchrun.err; tail -c 300 gpurun_out/r3t/bench_torchrun_forcesync.json
