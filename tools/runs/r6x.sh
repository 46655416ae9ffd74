#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/pmc_fc2.py > /dev/null 2> $O/pmc_$n.err
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "gemm256p" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], [round(x) for x in v])
PY
done
rm -rf $O/pmc_*/*/*.db
