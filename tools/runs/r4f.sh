#!/bin/bash
# round 4: bench lines of the side configurations (one box, back to back)
O=gpurun_out/r4f; mkdir -p $O
run() { n=$1; shift; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step "$@" > $O/$n.json 2> $O/$n.err; python3 -c "
import json,sys; d=json.loads(open('$O/$n.json').read().strip().split(chr(10))[-1]); print('$n', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', round(d['roofline']['frac'],4) if 'roofline' in d else '')"; }
run default
run force_gradsync --force-gradsync
run cu_hog16 --cu-hog 16
run cu_hog16_reserve16 --cu-hog 16 --reserve-cus 16
run default_again
run vitl --model vit_large
run tokens6400 --frames 32 --img-size 320 --batch 8
