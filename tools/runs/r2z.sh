set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2zz; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step > $O/bench_profiled.json 2> $O/prof.err
head -12 $O/prof/*/*kernel_stats.csv | cut -c1-140
