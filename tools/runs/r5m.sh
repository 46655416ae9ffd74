#!/bin/bash
# round 5: the three single-GPU configurations of BASELINE.json at the current commit, one box, plus the in-process A/B of the attention backward on ViT-L and 6400 tokens
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vitb.json 2> $O/bench_vitb.err
python3 bench.py --model vit_large --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_vitl.json 2> $O/bench_vitl.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_6400.json 2> $O/bench_6400.err
for f in vitb vitl 6400; do python3 -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', round(r['frac'],4), round(r.get('frac_of_sustained') or 0,4), d['peak_mem_gib'])"; done
timeout 900 python3 tools/ab_inproc.py --model vit_large attn_dkdv=0,1 > $O/ab_vitl.log 2>&1; tail -1 $O/ab_vitl.log
timeout 900 python3 tools/ab_inproc.py --frames 32 --img-size 320 --batch 8 attn_dkdv=0,1 > $O/ab_6400.log 2>&1; tail -1 $O/ab_6400.log
