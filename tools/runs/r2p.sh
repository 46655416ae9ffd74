set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step > $O/bench_profiled.json 2> $O/prof.err
for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$c.err; done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_MFMA -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_MFMA.err
find $O -name "*.csv" | head -20; du -sh $O
