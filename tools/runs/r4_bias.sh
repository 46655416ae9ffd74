#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4bias; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
DEVIAS_ATTN_BIAS_FUSED=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$m -- python3 $R/tools/exp/attn_bias_probe.py > /dev/null 2> $O/t$m.err
done
cd $R; for m in 0 1; do echo "== DEVIAS_ATTN_BIAS_FUSED=$m"; S=$(find $O/t$m -name "*kernel_stats.csv" | head -1); head -6 $S | cut -d, -f1-4 | cut -c1-160; done
rm -rf $O/t*/*/*.db
