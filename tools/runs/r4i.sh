#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_measured_path_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E" | head -8
run() { n=$1; shift; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step "$@" > $O/$n.json 2> $O/$n.err; python3 -c "
import json,sys; d=json.loads(open('$O/$n.json').read().strip().split(chr(10))[-1]); print('$n', round(d['value'],1), 'clips/s', round(d['ms_per_step'],3), 'ms')"; }
run a; run b; run c
