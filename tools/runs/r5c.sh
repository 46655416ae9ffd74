#!/bin/bash
# round 5: where the one-wave-per-SIMD dK / dV kernel spends its time: ablation builds (each leaves one ingredient out of the slice loop) + one PMC pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
rm -f $O/abl.log
for m in "" abl1 abl2 abl4 abl8 abl16 abl32 abl5 abl21 abl28 abl63; do
  if [ -z "$m" ]; then timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1
  else DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$m.so timeout 120 python3 tools/exp/dkdv1w_check.py timeonly >> $O/abl.log 2>&1; fi
done
grep "backward with" $O/abl.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python3 $R/tools/exp/dkdv1w_check.py timeonly > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- python3 $R/tools/exp/dkdv1w_check.py timeonly > $O/pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5c")
for d in ("pmc1", "pmc2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dkdv" in k or "bwd_dq" in k:
                acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(d, k, {n: sum(v) / len(v) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
    for f in glob.glob(os.path.join(O, d, "**", "*kernel_trace.csv"), recursive=True):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mhsa" in k:
                dur[k[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            v.sort(); print(d, "duration us median", round(v[len(v) // 2], 1), "n", len(v), k)
PY
rm -rf $O/pmc*/*/*.db 2>/dev/null; du -sh $O
