#!/bin/bash
# round 5: the one-wave-per-SIMD dK / dV kernel in the test suite and in the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mhsa" > $O/t_mhsa.log 2>&1; tail -3 $O/t_mhsa.log
timeout 900 python3 -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py -x -q > $O/t_regions.log 2>&1; tail -3 $O/t_regions.log
timeout 900 python3 -m pytest tests/test_measured_path_gpu.py -x -q > $O/t_measured.log 2>&1; tail -3 $O/t_measured.log
timeout 600 python3 tools/ab_inproc.py attn_dkdv=0,1 > $O/ab.log 2>&1; tail -2 $O/ab.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['full_step']['ms_per_step'])"
