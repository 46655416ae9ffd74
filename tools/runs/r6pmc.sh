#!/bin/bash
# the three PMC passes of tools/runs/r6fin.sh alone (after a host-side change to a kernel source file: the summary is keyed by the source hash), attention + regions tests, one default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r6pmc}; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_regions_gpu.py -x -q -k "mhsa or regions or prescale" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$n.err
done
cd $R; python3 tools/pmc_summary.py --json $O/pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_summary.txt; head -4 $O/pmc_summary.txt | cut -c1-200
rm -rf $O/pmc_*/*/*.db 2>/dev/null
cp $O/pmc_summary.json profiles/r6_pmc/summary.json        # (on the box only: lets the bench line below quote the traffic)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
