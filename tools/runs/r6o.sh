#!/bin/bash
# which kernels pay for 16 held CUs (reserve 16)?  kernel stats of bench.py with and without the hog, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes > $O/plain.json 2> $O/plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hog -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step --no-probes --cu-hog 16 --reserve-cus 16 > $O/hog.json 2> $O/hog.err
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r6o"
def load(d):
    f = glob.glob(O + "/" + d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("plain"), load("hog")
rows = sorted(a, key=lambda k: -(b.get(k, (0, 0))[1] - a[k][1]))
print("ms per step (15 traced steps), plain -> 16 CUs held + reserve 16, by kernel, largest increase first")
for k in rows[:16]:
    print(f"{a[k][1] / 15e6:8.3f} -> {b.get(k, (0, 0))[1] / 15e6:8.3f}  ({(b.get(k, (0, 0))[1] - a[k][1]) / 15e6:+.3f})  calls {a[k][0]:5d}  {k[:110]}")
print("sum", sum(v[1] for v in a.values()) / 15e6, "->", sum(v[1] for k, v in b.items() if "cu_hog" not in k) / 15e6)
PY
rm -rf $O/*/*/*.db
