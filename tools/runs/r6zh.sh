#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zh; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_regions_gpu.py -x -q -k "prescale" > $O/t.txt 2>&1; tail -15 $O/t.txt | cut -c1-300
