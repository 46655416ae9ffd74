#!/usr/bin/env python3
"""The K = 3072 / N = 768 forward shape (fc2: A = the GELU output [M, 3072], three column tiles per row panel) and, for comparison, qkv (K = 768, nine column tiles per panel)
for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: is the A panel fetched once per XCD group or once per column tile?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
h, u, x = bf(M, F), bf(M, D), bf(M, D)
W2, Wqkv = bf(D, F), bf(3 * D, D)
b2, bq = torch.randn(D, device="cuda") * 0.1, torch.randn(3 * D, device="cuda") * 0.1
for _ in range(3):
    o.gemm(h, W2, bias=b2, res=x)               # fc2 forward: algorithmic reads 308 (A) + 4.7 (W) + 77 (residual) MB, writes 77 MB
    o.gemm(u, Wqkv, bias=bq)                    # qkv forward: reads 77 + 3.5 MB, writes 231 MB
torch.cuda.synchronize()
