#!/bin/bash
# round 6, call 6: the whole GPU suite on the state of the commit, the bench line (with roofline.gemm_shapes), the kernel trace -> families + ledger
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_profiled.json 2> $O/trace.err
cd $R
python3 tools/gemm_ledger.py $O/trace profiles/r6_vendor_gemm_names.tsv > $O/gemm_shapes.txt 2>&1; cat $O/gemm_shapes.txt | grep -v "vendor kernel"
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv; python3 tools/kernel_families.py $S 15 > $O/kernel_families.txt; cat $O/kernel_families.txt
rm -rf $O/trace/*/*.db 2>/dev/null; du -sh $O
