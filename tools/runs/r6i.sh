#!/bin/bash
# round 6, call 10: LayerNorm backward with its rows interleaved over the grid ('' = new default) against the contiguous block per workgroup (_ln_blk), + 8 waves x 2 / 3 rows ahead
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
for v in "" _ln_blk _ln_c _ln_d "" _ln_blk; do
  L=$R/devias_amd/libdevias_amd.so; [ -n "$v" ] && L=$R/tools/exp/libdevias_amd$v.so
  echo "== LN variant '$v'" >> $O/ln.txt
  DEVIAS_LIB_PATH=$L timeout 120 python3 tools/exp/ln_ab.py 2>&1 | grep -v amdgpu.ids >> $O/ln.txt
done
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" 2>&1 | tail -1 >> $O/ln.txt
timeout 600 python3 -m pytest tests/test_regions_gpu.py -x -q 2>&1 | tail -1 >> $O/ln.txt
cat $O/ln.txt
timeout 300 python3 tools/ab_inproc.py gemm_tail_split=2,3 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
