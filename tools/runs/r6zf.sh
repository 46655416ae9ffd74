#!/bin/bash
# LayerNorm backward at D = 1024 (ViT-L): waves per workgroup x rows in flight ahead (variant builds), alone, processes interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zf; mkdir -p $O
cd $R
for rep in 1 2; do for v in l4a l4b l4c l4d; do echo "== $v" >> $O/lnb4.txt; DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$v.so timeout 120 python3 tools/exp/lnb4_time.py 2>&1 | grep -E "ln_bwd|Error|error" | tail -2 >> $O/lnb4.txt; done; done; cat $O/lnb4.txt
