#!/bin/bash
# round 3, call n: board power and shader clock while the bench loop runs (is the step power-capped?)
mkdir -p gpurun_out/r3n
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 3000; echo; sleep 0.5; done) > gpurun_out/r3n/smi.txt 2>&1 &
SMI=$!
python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-full-step > gpurun_out/r3n/bench.json 2>/dev/null
wait $SMI
rocm-smi --showmaxpower 2>/dev/null | tail -5 > gpurun_out/r3n/maxpower.txt
tail -c 1500 gpurun_out/r3n/smi.txt; cat gpurun_out/r3n/maxpower.txt
