#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4dq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pm -- python3 $R/tools/exp/attn_bwd_probe.py > /dev/null 2> $O/pm.err
cd $R; python3 tools/pmc_summary.py $O/pm | grep -A1 "mhsa_bwd" | cut -c1-400
rm -rf $O/pm/*/*.db
