#!/bin/bash
# round 3, call y: PMC passes over tools/pmc_probe.py, final code (tail split, stream-K policy off, 16-wave LayerNorm backward)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_$n.err
done
cd $R; python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/summary.txt; head -40 $O/summary.txt
