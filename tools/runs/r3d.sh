R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python tools/attn_cfg.py DEVIAS_ATTN_CFG=0 DEVIAS_ATTN_CFG=6 DEVIAS_ATTN_CFG=7 DEVIAS_ATTN_CFG=0 DEVIAS_ATTN_CFG=6 2>&1 | grep -v amdgpu.ids
DEVIAS_ATTN_CFG=6 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mhsa" 2>&1 | tail -5
