#!/usr/bin/env python3
"""ms per step by kernel family from a rocprofv3 --kernel-trace --stats CSV of `bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-step`
(15 steps run under the profiler: 2 warm-up + 8 timed + 5 host-cost steps).  usage: kernel_families.py <kernel_stats.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
fam = [("GEMM", ("gemm256", "gemm_kernel", "gemm_ss", "gemm_pk", "gemm_smallm")), ("attention backward dK/dV", ("mhsa_bwd_dkdv",)), ("attention backward dQ", ("mhsa_bwd_dq",)),
       ("attention forward", ("mhsa_fwd",)), ("LayerNorm fwd + bwd", ("ln_fwd", "ln_bwd")), ("split-K reduces", ("splitk_reduce",)),
       ("column sums", ("colsum",)), ("slot attention", ("slotm", "slotf", "slot_")), ("LayerNorm parameter reduce", ("ln_param_reduce",))]
tot = {k: 0.0 for k, _ in fam}; tot["rest"] = 0.0
calls = 0
for r in rows:
    t = float(r["TotalDurationNs"]) / steps / 1e6; calls += int(r["Calls"])
    for k, pats in fam:
        if any(p in r["Name"] for p in pats): tot[k] += t; break
    else: tot["rest"] += t
s = sum(tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"{k:32s} {v:7.2f} ms/step  {100 * v / s:5.1f} %")
print(f"{'sum':32s} {s:7.2f} ms/step; {calls / steps:.0f} kernel launches per step")
