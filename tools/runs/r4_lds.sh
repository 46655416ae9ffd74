#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/p1.err; tail -2 $O/p1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/p2 -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/p2.err; tail -2 $O/p2.err
cd $R; python3 tools/pmc_summary.py $O/p1 $O/p2 > $O/summary.txt; cut -c1-500 $O/summary.txt | head -40
rm -rf $O/p*/*/*.db
