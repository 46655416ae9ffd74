#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final_tests3; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
