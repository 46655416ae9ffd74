#!/usr/bin/env python3
"""Time of the block's forward GEMMs against the number of tile ROUNDS (M chosen so that the 256x256 tiles fill r rounds of 256 CUs almost
exactly): per-round cost and the intercept (what a launch costs beyond its rounds), for this library's kernels with their real epilogues and
for torch.matmul (vendor library; calibration only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from devias_amd import ops as o
from tools.microbench import timeit

dev = "cuda"
bf = torch.bfloat16
ROUNDS = (1, 2, 3, 4, 6, 8, 10, 12)
vendor = "--vendor" in sys.argv
for name, N, K, epi in (("qkv bias", 2304, 768, "bias"), ("fc1 bias", 3072, 768, "bias"), ("fc1 bias+gelu+aux", 3072, 768, "gelu"),
                        ("proj bias+res", 768, 768, "res"), ("fc2 bias+res", 768, 3072, "res")):
    ncol = N // 256
    ts, rs = [], []
    row = []
    for r in ROUNDS:
        mt = (256 * r) // ncol
        M = mt * 256
        a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf)
        bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=bf)
        aux = torch.empty(M, N, device=dev, dtype=bf) if epi == "gelu" else None
        res = torch.randn(M, N, device=dev).to(bf) if epi == "res" else None
        if vendor:
            fn = lambda: torch.matmul(a, w.t(), out=out)
        elif epi == "gelu":
            fn = lambda: o.gemm(a, w, bias=bias, act=o.ACT_GELU, aux_out=aux, out=out)
        elif epi == "res":
            fn = lambda: o.gemm(a, w, bias=bias, res=res, out=out)
        else:
            fn = lambda: o.gemm(a, w, bias=bias, out=out)
        t = timeit(fn, iters=20) * 1e3
        rounds = mt * ncol / 256.0
        ts.append(t); rs.append(rounds)
        row.append(f"{rounds:5.2f}r:{t:6.1f}")
    slope, icpt = np.polyfit(rs, ts, 1)
    print(f"{name:20s} " + "  ".join(row) + f"   -> {slope:5.2f} us/round + {icpt:5.1f} us" + ("  (vendor)" if vendor else ""))
    sys.stdout.flush()
