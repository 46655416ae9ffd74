#!/bin/bash
# round 6, call 11: the other BASELINE configurations on the round's code: bench lines + in-process A/B of the round's options there
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_vitb.json 2> $O/bench_vitb.err
python3 bench.py --model vit_large --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_vitl.json 2> $O/bench_vitl.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_6400.json 2> $O/bench_6400.err
for f in vitb vitl 6400; do python3 -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', round(r['frac'],4), round(r.get('frac_of_sustained') or 0,4))"; done
timeout 900 python3 tools/ab_inproc.py --model vit_large gemm_wt=0,1 gemm_epi_spec=0,1 gemm_tail_split=2,3 > $O/ab_vitl.txt 2>&1; grep -v amdgpu.ids $O/ab_vitl.txt
timeout 900 python3 tools/ab_inproc.py --frames 32 --img-size 320 --batch 8 gemm_wt=0,1 gemm_epi_spec=0,1 gemm_tail_split=2,3 > $O/ab_6400.txt 2>&1; grep -v amdgpu.ids $O/ab_6400.txt
