#!/bin/bash
# round 5, first call: the new measurement keys of bench.py + the tests touched by the housekeeping commit
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
timeout 900 python3 -m pytest tests/test_regions_gpu.py tests/test_kernels_gpu.py -x -q -k "defer or dynamic or queue or persistent" > $O/tests.log 2>&1; tail -5 $O/tests.log
