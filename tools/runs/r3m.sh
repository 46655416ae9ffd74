#!/bin/bash
# round 3, call m: in-step kernel durations, eight-wave policy (W4=0) against the four-wave kernels (W4=15), same box
mkdir -p gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp
for w in 0 3; do
  export DEVIAS_GEMM_W4=$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3m/prof_w$w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3m/bench_profiled_w$w.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3m/prof_w$w.err
done
