#!/bin/bash
# round 6, call 9: LayerNorm backward ablations (diagnostic builds, results wrong on purpose): where do its 53 us go?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O
cd $R
for v in "" _ln_a1 _ln_a2 _ln_a3 _ln_a4 _ln_c _ln_ca1; do
  L=$R/devias_amd/libdevias_amd.so; [ -n "$v" ] && L=$R/tools/exp/libdevias_amd$v.so
  echo "== LN variant '$v'" >> $O/ln_ablate.txt
  DEVIAS_LIB_PATH=$L timeout 120 python3 tools/exp/ln_ab.py 2>&1 | grep -v amdgpu.ids >> $O/ln_ablate.txt
done; cat $O/ln_ablate.txt
