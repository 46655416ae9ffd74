#!/bin/bash
# round 6, final state: the whole GPU suite, bench lines (ViT-B + the two side configurations + the CU-hog pair), rocprofv3 kernel trace + stats, PMC passes (JSON keyed by the kernel-source hash)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r6fin3}; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -2
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
python3 bench.py --model vit_large --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_vitl.json 2> $O/bench_vitl.err
python3 bench.py --frames 32 --img-size 320 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $O/bench_6400.json 2> $O/bench_6400.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --cu-hog 16 > $O/bench_hog16.json 2> $O/bench_hog16.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --cu-hog 16 --reserve-cus 16 > $O/bench_hog16_res16.json 2> $O/bench_hog16_res16.err
for f in bench bench_vitl bench_6400 bench_hog16 bench_hog16_res16; do python3 -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', round(r['frac'],4), round(r.get('frac_of_sustained') or 0,4))"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_profiled.json 2> $O/trace.err
cd $R
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
cp $S $O/kernel_stats.csv; python3 tools/kernel_families.py $S 15 > $O/kernel_families.txt; cat $O/kernel_families.txt
python3 tools/trace_gaps.py $T 15 > $O/trace_gaps.txt; tail -3 $O/trace_gaps.txt
python3 tools/gemm_ledger.py $O/trace profiles/r6_vendor_gemm_names.tsv --json $O/gemm_shapes.json > $O/gemm_shapes.txt 2>&1; grep -v "vendor kernel" $O/gemm_shapes.txt
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$n.err
done
cd $R; python3 tools/pmc_summary.py --json $O/pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_summary.txt; head -14 $O/pmc_summary.txt
rm -rf $O/trace/*/*.db $O/pmc_*/*/*.db 2>/dev/null; du -sh $O
