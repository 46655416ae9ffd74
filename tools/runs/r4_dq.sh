#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4dq; mkdir -p $O
cd $R; python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k mhsa 2>&1 | tail -1
python tools/exp/attn_bwd_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tools/exp/attn_bwd_probe.py > /dev/null 2> $O/tr.err
S=$(find $O/tr -name "*kernel_stats.csv" | head -1); head -4 $S | cut -c1-200
rm -rf $O/tr/*/*.db
cd $R; python tools/ab_inproc.py gemm_debug=0,0 2>&1 | tail -1
