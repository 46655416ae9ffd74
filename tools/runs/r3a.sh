set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_regions_gpu.py -x -q -m gpu > $O/regions.log 2>&1; tail -15 $O/regions.log
timeout 600 python bench.py --no-cpu-baseline --no-full-step > $O/bench_regions.json 2> $O/bench.err; tail -c 900 $O/bench_regions.json
DEVIAS_REGIONS=0 timeout 600 python bench.py --no-cpu-baseline --no-full-step > $O/bench_noregions.json 2>> $O/bench.err; tail -c 900 $O/bench_noregions.json
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
