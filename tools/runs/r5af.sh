#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5af; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_regions_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "layernorm or ln or agg or region or parity or golden" > $O/t.log 2>&1; tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench.json 2> $O/err.txt
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
grep "ln_bwd\|ln_fwd\|smallm" $S | cut -c1-160
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
rm -rf $O/trace/*/*.db
