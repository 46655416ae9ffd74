#!/bin/bash
# round 6, call 1: the specialised-epilogue build (gemm_epi_spec) -- correctness of the persistent kernels, in-process A/B against the generic form, the per-shape
# ledger at HEAD beside the vendor library's times and kernel names, per-tile stamps of the block's shapes, the --cu-hog lines (item 6)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or gemm_epilogues or gemm_layouts or ring_belongs" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python3 tools/ab_inproc.py gemm_epi_spec=0,1 > $O/ab_epi_spec.txt 2>&1; cat $O/ab_epi_spec.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_profiled.json 2> $O/trace.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/vendor -- python3 $R/tools/vendor_gemm_names.py run > /dev/null 2> $O/vendor.err
cd $R
python3 tools/vendor_gemm_names.py parse $O/vendor > $O/vendor.tsv 2>> $O/vendor.err; cat $O/vendor.tsv
python3 tools/gemm_ledger.py $O/trace $O/vendor.tsv --json $O/gemm_shapes.json > $O/gemm_shapes.txt 2>&1; cat $O/gemm_shapes.txt
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv; python3 tools/kernel_families.py $S 10 > $O/kernel_families.txt; cat $O/kernel_families.txt
# per-tile stamps (debug build), eight-wave persistent kernel
for s in qkv fc1g proj fc2 dfc2 dfc1 dproj dqkv; do timeout 120 python3 tools/gemm_pstamps.py $s >> $O/pstamps.txt 2>&1; done
echo "== generic epilogue (gemm_epi_spec = 0)" >> $O/pstamps.txt
for s in qkv fc1g proj dfc2; do DEVIAS_GEMM_EPI_SPEC=0 timeout 120 python3 tools/gemm_pstamps.py $s >> $O/pstamps.txt 2>&1; done; cat $O/pstamps.txt
# co-residency: 16 CUs held during every backward, with and without the reserve
timeout 300 python3 tools/ab_inproc.py hog=16 gemm_reserve_cus=0,16 > $O/cu_hog.txt 2>&1; cat $O/cu_hog.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --cu-hog 16 > $O/bench_hog16.json 2> $O/bench_hog16.err; tail -c 300 $O/bench_hog16.json
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --cu-hog 16 --reserve-cus 16 > $O/bench_hog16_res16.json 2> $O/bench_hog16_res16.err; tail -c 300 $O/bench_hog16_res16.json
# LayerNorm backward variants (waves per workgroup, rows in flight ahead): time + correctness of each
for v in "" _ln_c _ln_d _ln_e _ln_i; do
  L=$R/devias_amd/libdevias_amd.so; [ -n "$v" ] && L=$R/tools/exp/libdevias_amd$v.so
  echo "== LN variant '$v'" >> $O/ln_variants.txt
  DEVIAS_LIB_PATH=$L timeout 120 python3 tools/exp/ln_ab.py >> $O/ln_variants.txt 2>&1
  DEVIAS_LIB_PATH=$L timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" 2>&1 | tail -1 >> $O/ln_variants.txt
done; cat $O/ln_variants.txt
rm -rf $O/trace/*/*.db $O/vendor/*/*.db 2>/dev/null; du -sh $O
