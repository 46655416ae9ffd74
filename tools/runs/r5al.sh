#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5al; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_measured_path_gpu.py -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench.json 2> $O/err.txt
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
grep "ln_bwd\|reduce_jobs" $S | cut -c1-70,120-200
rm -rf $O/trace/*/*.db
