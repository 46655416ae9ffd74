#!/usr/bin/env python3
"""In-step durations of the encoder block's forward / dgrad GEMMs from two rocprofv3 kernel traces of bench.py (same box): the kernels of a block run in a
fixed order, so the launches of one kernel template cycle through the shapes it serves.  usage: step_gemm_shapes.py <trace A dir> [<trace B dir>]"""
import csv, glob, statistics as st, sys

def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows

def seq(rows, key):
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]

def med(v): return st.median(v) if v else float("nan")

def shapes(rows):
    """per shape: median in-step duration; which kernel template serves a shape follows from the names present in the trace"""
    out = {}
    def has(k): return any(k in r["Kernel_Name"] for r in rows)
    # forward, no side rows: qkv, fc1 alternate on one template
    k = "gemm256w_kernel<false, 0>" if has("gemm256w_kernel<false, 0>") else "gemm256p_kernel<false, 0>"
    d = seq(rows, k); out["qkv"], out["fc1+gelu"] = med(d[0::2]), med(d[1::2])
    # forward + residual: proj, fc2 (the patch embedding's launch comes first in every step)
    if has("gemm256w_kernel<false, 1>"):
        d = seq(rows, "gemm256w_kernel<false, 1>")[1:]; out["proj+res"], out["fc2+res"] = med(d[0::2]), med(d[1::2])
    elif has("gemm256sk_kernel<false, 1>"):
        out["proj+res"] = med(seq(rows, "gemm256p_kernel<false, 1>")); out["fc2+res"] = med(seq(rows, "gemm256sk_kernel<false, 1>"))
    else:
        d = seq(rows, "gemm256p_kernel<false, 1>")[1:]; out["proj+res"], out["fc2+res"] = med(d[0::2]), med(d[1::2])
    # dgrad, no side rows: dfc1, dproj, dqkv
    if has("gemm256w_kernel<true, 0>"):
        d = seq(rows, "gemm256w_kernel<true, 0>"); out["dfc1"], out["dproj"], out["dqkv"] = med(d[0::3]), med(d[1::3]), med(d[2::3])
    elif has("gemm256sk_kernel<true, 0>"):
        d = seq(rows, "gemm256sk_kernel<true, 0>"); out["dfc1"], out["dqkv"] = med(d[0::2]), med(d[1::2])
        out["dproj"] = med(seq(rows, "gemm256p_kernel<true, 0>"))
    else:
        d = seq(rows, "gemm256p_kernel<true, 0>"); out["dfc1"], out["dproj"], out["dqkv"] = med(d[0::3]), med(d[1::3]), med(d[2::3])
    out["dfc2+dgelu"] = med(seq(rows, "gemm256w_kernel<true, 2>" if has("gemm256w_kernel<true, 2>") else "gemm256p_kernel<true, 2>"))
    out["wgrad (mean of 4)"] = med(seq(rows, "gemm256_kernel<true, true"))
    out["attn fwd"] = med(seq(rows, "mhsa_fwd32")); out["attn dq"] = med(seq(rows, "mhsa_bwd_dq")); out["attn dkdv"] = med(seq(rows, "mhsa_bwd_dkdv"))
    return out

a = shapes(load(sys.argv[1]))
b = shapes(load(sys.argv[2])) if len(sys.argv) > 2 else a
ta = tb = 0.0
for k in a:
    print(f"{k:20s} {a[k]:8.1f} us   {b[k]:8.1f} us   {b[k] - a[k]:+7.1f}")
    if k in ("dfc1", "dproj", "dqkv", "qkv", "fc1+gelu", "proj+res", "fc2+res", "dfc2+dgelu"): ta += a[k]; tb += b[k]
print(f"{'block fwd + dgrad':20s} {ta:8.1f} us   {tb:8.1f} us   {tb - ta:+7.1f}")
