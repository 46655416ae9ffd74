#!/usr/bin/env python3
"""Summarise separate rocprofv3 --pmc passes over tools/pmc_probe.py: per kernel, the median of each counter over its launches, and the HBM-side traffic
(2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md).
usage: pmc_summary.py [--json OUT.json] <pass dir> [<pass dir> ...]
--json also writes the per-kernel medians with the hash of the kernel sources the profiled library was built from (devias_amd.build.source_hash):
bench.py quotes `traffic` from that file only while the hash still matches the sources it runs."""
import csv, glob, json, os, statistics as st, sys, collections
json_out = None
if "--json" in sys.argv:
    i = sys.argv.index("--json"); json_out = sys.argv[i + 1]; del sys.argv[i:i + 2]
summary = {}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, k), cs in per.items():
            for c, v in cs.items(): vals[k][c].append(v)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals, key=lambda k: -st.median(durs.get(k, [0]))):
    if "devias" not in k and "GLOBAL__N" not in k and "anonymous" not in k: continue
    m = {c: st.median(v) for c, v in vals[k].items()}
    du = st.median(durs[k]) if durs.get(k) else float("nan")
    print(k[:110])
    print("    " + "  ".join(f"{c}={m[c]:.4g}" for c in sorted(m)) + f"  dur_us={du:.4g}")
    entry = dict(m); entry["dur_us"] = du; entry["launches"] = max((len(v) for v in vals[k].values()), default=0)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        t = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        entry["traffic_bytes"] = t
        print(f"    traffic = {t / 1e6:.1f} MB per launch over {du:.1f} us = {t / 1e6 / du:.3f} MB/us (= TB/s)")
    summary[k] = entry
if json_out:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from devias_amd import build
    json.dump({"source_hash": build.source_hash(), "probe": "tools/pmc_probe.py", "method": "rocprofv3 --kernel-trace --pmc, separate passes per counter group; "
               "median over the launches of a kernel; traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction)", "kernels": summary},
              open(json_out, "w"), indent=1)
    print("wrote", json_out)
