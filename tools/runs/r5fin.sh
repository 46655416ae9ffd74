#!/bin/bash
# round 5, final code: profile (bench, trace, PMC) and the two side configurations, one gpurun call
R=$GRAFT_REPO_ROOT
bash $R/tools/runs/r5_profile.sh
O=$R/gpurun_out/r5fin4; cd $R
python3 bench.py --model vit_large --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_vitl.json 2> $O/bench_vitl.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_6400.json 2> $O/bench_6400.err
for f in vitl 6400; do python3 -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', round(r['frac'],4), round(r.get('frac_of_sustained') or 0,4))"; done
