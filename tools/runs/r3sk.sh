#!/bin/bash
# round 3: in-step per-shape durations with the stream-K policy on / off (tail split on), same box
mkdir -p gpurun_out/r3sk
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export DEVIAS_GEMM_SK=$v; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3sk/prof_sk$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3sk/bench_sk$v.json 2>/dev/null
done
cd $GRAFT_REPO_ROOT; python3 tools/step_gemm_shapes.py gpurun_out/r3sk/prof_sk1 gpurun_out/r3sk/prof_sk0
