#!/bin/bash
# round 5: bench lines, rocprofv3 kernel trace + stats, PMC passes (JSON keyed by the kernel-source hash), on the code as committed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5fin4; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_profiled.json 2> $O/trace.err
cd $R
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
cp $S $O/kernel_stats.csv; python3 tools/kernel_families.py $S 15 > $O/kernel_families.txt; cat $O/kernel_families.txt
python3 tools/trace_gaps.py $T 15 > $O/trace_gaps.txt; cat $O/trace_gaps.txt
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$n.err
done
cd $R; python3 tools/pmc_summary.py --json $O/pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_summary.txt; head -12 $O/pmc_summary.txt
rm -rf $O/trace/*/*.db $O/pmc_*/*/*.db 2>/dev/null; du -sh $O
