#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad or gemm" 2>&1 | tail -1
for v in 0 1 0 1; do echo "== gemm_splitk_xcd $v"; DEVIAS_GEMM_SPLITK_XCD=$v python tools/exp/wgrad_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-70; done
python tools/ab_inproc.py gemm_splitk_xcd=0,1 2>&1 | tail -1
python tools/ab_inproc.py gemm_splitk_xcd=0,1 2>&1 | tail -1
