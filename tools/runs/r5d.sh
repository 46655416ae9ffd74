#!/bin/bash
# round 5: dK / dV one-wave kernel, v3 (hand-counted LDS waits): correctness, then ablation timings, then per-kernel durations under the kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
timeout 300 python3 tools/exp/dkdv1w_check.py > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log; grep -c "^ok" $O/check.log; grep "FAIL\|rc=" $O/check.log | cut -c1-250
rm -f $O/abl.log
for m in "" abl1 abl4 abl8 abl16 abl5 abl28 abl63; do
  if [ -z "$m" ]; then timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
  else DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$m.so timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1; fi
done
grep "backward with" $O/abl.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/exp/dkdv1w_check.py timeonly > $O/kt.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5d")
for f in glob.glob(os.path.join(O, "kt", "**", "*kernel_trace.csv"), recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mhsa" in k or "colsum" in k:
            dur[k[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        v.sort(); print("duration us median", round(v[len(v) // 2], 1), "n", len(v), k)
PY
rm -rf $O/kt/*/*.db 2>/dev/null
