set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --force-gradsync > $O/bench_forcesync.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --cu-hog 16 > $O/bench_cuhog16.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --cu-hog 16 --reserve-cus 16 > $O/bench_cuhog16_reserve16.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --reserve-cus 16 > $O/bench_reserve16.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-full-step --cu-hog 32 --reserve-cus 32 > $O/bench_cuhog32_reserve32.json 2>> $O/bench.err
timeout 900 python bench.py --model vit_large --no-full-step --cpu-steps 1 > $O/vitl_bench.json 2>> $O/bench.err
timeout 900 python bench.py --frames 32 --img-size 320 --batch 8 --no-full-step --cpu-steps 1 > $O/6400_bench.json 2>> $O/bench.err
timeout 600 python tools/host_cost.py > $O/host_cost.txt 2>&1
timeout 600 python tools/vendor_gemm_ref.py > $O/vendor_gemm_calibration.txt 2>&1
timeout 600 python tools/gemm_block_shapes.py "" > $O/gemm_block_shapes.txt 2>&1
timeout 600 python tools/attn_cfg.py DEVIAS_ATTN_CFG=0 DEVIAS_ATTN_CFG=6 DEVIAS_ATTN_CFG=0 DEVIAS_ATTN_CFG=6 > $O/attn_fwd_ab.txt 2>&1
grep -h -o '"value": [0-9.]*, "unit": "clips/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/*.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step > $O/bench_profiled.json 2> $O/prof.err
export DEVIAS_ROCTX=1
timeout 600 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/prof_markers -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step > $O/bench_markers.json 2> $O/prof_markers.err
unset DEVIAS_ROCTX
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$n.err
done
ls $O $O/prof/* $O/prof_markers/* | head -60
