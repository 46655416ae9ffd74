#!/bin/bash
# round 3, call l: kernel stats of the step with the four-wave GEMM default
mkdir -p gpurun_out/r3l
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3l/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $GRAFT_REPO_ROOT/gpurun_out/r3l/bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3l/prof.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3l/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/15/1e6:7.3f} ms/step {int(r['Calls'])/15:6.1f} calls avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
