#!/bin/bash
# round 5 checkpoint: the whole GPU suite, smoke, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_of_sustained'], d['full_step']['ms_per_step'], r['dominant_kernel']['avg_ms'], r['probe_fc1_fwd']['avg_ms'])"
