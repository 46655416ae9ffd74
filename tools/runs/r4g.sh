#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_regions_gpu.py tests/test_parity_gpu.py tests/test_parallel_cpu.py -x -q -k "not bench_launches" 2>&1 | tail -3
python tools/exp/gradsync_copies.py 2>&1 | grep -v amdgpu | tail -3
run() { n=$1; shift; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step "$@" > $O/$n.json 2> $O/$n.err; python3 -c "
import json,sys; d=json.loads(open('$O/$n.json').read().strip().split(chr(10))[-1]); print('$n', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', 'host idle', round(d['host_idle_enqueue_ms_per_step'],2))"; }
run default
run force_gradsync --force-gradsync
run default_again
run force_gradsync_again --force-gradsync
