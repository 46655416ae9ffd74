#!/usr/bin/env python3
"""Idle time between consecutive kernels of the step from a rocprofv3 --kernel-trace CSV (Start/End timestamps): how much of a step the GPU spends with NO kernel
running (dispatch latency between dependent launches), and which successor kernels sit behind the longest gaps.  usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv, sys, collections, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
busy_end = ev[0][1]
gaps = []
per = collections.defaultdict(list)
for s, e, n in ev[1:]:
    g = s - busy_end
    if 0 < g < 200000:                    # (gaps above 200 us are between steps / host stalls, not dispatch latency)
        gaps.append(g); per[n.split("(")[0][-60:]].append(g)
    busy_end = max(busy_end, e)
tot = sum(gaps)
print(f"{len(ev)} kernels, {len(gaps)} gaps below 200 us: {tot / 1e6 / steps:.3f} ms per step idle between kernels; median gap {st.median(gaps) / 1e3:.2f} us, p90 {sorted(gaps)[int(0.9 * len(gaps))] / 1e3:.2f} us")
for n, g in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {sum(g) / 1e6 / steps:7.3f} ms/step  {len(g) / steps:6.1f} gaps/step  median {st.median(g) / 1e3:5.2f} us   before {n}")
