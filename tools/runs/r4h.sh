#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default force; do
  extra=""; [ $v = force ] && extra="--force-gradsync"
  DEVIAS_GEMM_DYNAMIC=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step $extra > $O/bench_$v.json 2> $O/trace_$v.err
  S=$(find $O/trace_$v -name "*kernel_stats.csv" | head -1); T=$(find $O/trace_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python3 $R/tools/kernel_families.py $S 15 | tail -4; python3 $R/tools/trace_gaps.py $T 15 | head -8
  python3 -c "
import json; d=json.loads(open('$O/bench_$v.json').read().strip().split(chr(10))[-1]); print('ms_per_step', round(d['ms_per_step'],2))"
  rm -rf $O/trace_$v/*/*.db
done
