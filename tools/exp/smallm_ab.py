#!/usr/bin/env python3
"""small-M GEMM kernel vs the split-K pair it replaces, alternating timing"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
for M, N, K in ((64, 3072, 768), (64, 768, 3072), (64, 768, 768), (96, 3072, 768), (64, 400, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ts = {0: [], 1: []}
    for rnd in range(6):
        for m in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            o.set_option("gemm_smallm", m)
            ts[m].append(timeit(lambda: o.gemm(a, w, bias=b, out=out), iters=30, warmup=3) * 1e3)
    print(f"M={M} N={N} K={K}: split-K pair {sorted(ts[0])[3]:6.1f} us   small-M kernel {sorted(ts[1])[3]:6.1f} us", flush=True)
