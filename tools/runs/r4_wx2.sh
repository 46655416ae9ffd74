#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4wx; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do for c in FETCH_SIZE "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do n=$(echo $c | cut -d' ' -f1)
DEVIAS_GEMM_SPLITK_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p${v}_$n -- python3 $R/tools/exp/wgrad_probe.py > /dev/null 2> $O/p${v}_$n.err
done; done
cd $R; for v in 0 1; do echo "== gemm_splitk_xcd $v"; python3 tools/pmc_summary.py $O/p${v}_FETCH_SIZE $O/p${v}_TCC_HIT | grep -A2 "gemm256_kernel<true, true" | cut -c1-300; done
rm -rf $O/p*/*/*.db
