#!/bin/bash
# round 5: the one-wave-per-SIMD dQ kernel: correctness beside the old kernels and fp32, the attention tests, per-kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
timeout 300 python3 tools/exp/dkdv1w_check.py > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log; grep -c "^ok" $O/check.log; grep "FAIL\|rc=\|Error\|error" $O/check.log | cut -c1-330 | head -8
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mhsa" > $O/t_mhsa.log 2>&1; tail -3 $O/t_mhsa.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/exp/dkdv1w_check.py timeonly > $O/kt.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5k")
for f in glob.glob(os.path.join(O, "kt", "**", "*kernel_trace.csv"), recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mhsa" in k:
            dur[k[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        v.sort(); print("duration us median", round(v[len(v) // 2], 1), "n", len(v), k)
PY
rm -rf $O/kt/*/*.db 2>/dev/null
