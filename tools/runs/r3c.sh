set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mhsa" > $O/mhsa_tests.log 2>&1; tail -5 $O/mhsa_tests.log
timeout 600 python tools/attn_cfg.py DEVIAS_ATTN_CFG=0 2>&1 | tail -3
timeout 600 python tools/attn_dkdv_ab.py 0,42 2>&1 | tail -3
