set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_recipe_gpu.py tests/test_regions_gpu.py "tests/test_kernels_gpu.py::test_head_match_loss" tests/test_measured_path_gpu.py::test_bf16_full_size_step_beside_the_oracle -x -q -m gpu > $O/new_tests.log 2>&1; tail -15 $O/new_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
for k in 8 16 32; do timeout 600 python bench.py --no-cpu-baseline --no-full-step --cu-hog $k > $O/bench_cuhog$k.json 2>> $O/bench.err; done
timeout 600 python bench.py --no-cpu-baseline --no-full-step --force-gradsync > $O/bench_forcesync.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --force-gradsync --cu-hog 16 > $O/bench_forcesync_cuhog16.json 2>> $O/bench.err
timeout 900 python bench.py --model vit_large --no-full-step --cpu-steps 1 > $O/vitl_bench.json 2>> $O/bench.err
timeout 900 python bench.py --frames 32 --img-size 320 --batch 8 --no-full-step --cpu-steps 1 > $O/6400_bench.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*, "unit": "clips/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/*.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $O/bench_profiled.json 2> $O/prof.err
ls $O/prof/*/ | head; head -14 $O/prof/*/*kernel_stats.csv | cut -c1-150
