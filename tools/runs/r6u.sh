#!/bin/bash
# cache-policy bits on the pinned K-tile's LDS-DMA (variant builds): A non-temporal, B non-temporal, A sc1; the block's GEMMs alone, then bench.py round-robin
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6u; mkdir -p $O
cd $R
timeout 600 python3 tools/gemm_block_shapes.py "" DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_ant.so DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_bnt.so DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_asc1.so "" 2>&1 | grep -v amdgpu.ids | grep -E "==|fwd|total" > $O/shapes.txt; cat $O/shapes.txt
timeout 900 python3 tools/ab_bench.py 3 "base:" "ant:DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_ant.so" "bnt:DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_bnt.so" "asc1:DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_asc1.so" > $O/ab.txt 2>&1; cat $O/ab.txt
