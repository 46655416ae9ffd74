#!/usr/bin/env python3
"""weight gradients of the agg block: [Nout, Kin] = dY^T X with a reduction over only R = 64 / 96 rows: 256x256 kernel (36 workgroups) vs 128x128 kernel (144)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
for R in (64, 96, 128, 256):
    for n, k in ((768, 3072), (3072, 768), (768, 768)):
        dy = torch.randn(R, n, device="cuda").bfloat16(); x = torch.randn(R, k, device="cuda").bfloat16()
        row = []
        for big in (1, 0):
            o.set_option("gemm256", big)
            o.counters(reset=True)
            w = o.wgrad(dy, x)
            c = {kk: v for kk, v in o.counters().items() if v}
            t = timeit(lambda: o.wgrad(dy, x), iters=50) * 1e3
            row.append(f"gemm256={big}: {t:6.1f} us {c}")
        print(f"R={R:4d} [{n},{k}]  " + "   ".join(row), flush=True)
