#!/bin/bash
# is the K = 3072 shapes' slower K-iteration (1.8 us against 1.6) the A operand's way from HBM?  fc2 / dfc1-shaped launches with every A row aliased to row 0 (lda = 0) against the real stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6r; mkdir -p $O
cd $R
for s in fc2 dfc1 qkv; do for a in "" 1; do echo "== $s PSTAMP_LDA0=$a" >> $O/pstamps.txt; PSTAMP_LDA0=$a timeout 120 python3 tools/gemm_pstamps.py $s 2>&1 | grep -E "remaining|back-to-back|first K" >> $O/pstamps.txt; done; done; cat $O/pstamps.txt | cut -c1-220
