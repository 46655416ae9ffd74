#!/bin/bash
# kernel trace of the ViT-L step (config 4) and of the 6400-token step (config 5): families and the top kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ze; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_vitl -- python3 $R/bench.py --model vit_large --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_vitl.json 2> $O/trace_vitl.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_6400 -- python3 $R/bench.py --frames 32 --img-size 320 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/bench_6400.json 2> $O/trace_6400.err
cd $R
for c in vitl 6400; do S=$(find $O/trace_$c -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats_$c.csv; python3 tools/kernel_families.py $S 13 > $O/kernel_families_$c.txt; echo "== $c"; cat $O/kernel_families_$c.txt; done
rm -rf $O/trace_*/*/*.db
