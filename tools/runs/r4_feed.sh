#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4feed; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
[ -n "$ONLY" ] || ONLY="1 2 3 4 5"
for c in "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_NC_READ_REQ TCP_TCC_UC_READ_REQ"; do
  i=$((i+1)); case " $ONLY " in *" $i "*) ;; *) continue;; esac
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pass$i -- python3 $R/tools/pmc_feed.py > /dev/null 2> $O/pass$i.err; tail -2 $O/pass$i.err
done
cd $R; python3 tools/pmc_summary.py $O/pass1 $O/pass2 $O/pass3 $O/pass4 $O/pass5 > $O/summary.txt; cat $O/summary.txt | cut -c1-400
rm -rf $O/pass*/*/*.db
