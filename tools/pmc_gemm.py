#!/usr/bin/env python3
"""A few launches of the persistent GEMM kernels at the fc1 shapes (M = 50176), for rocprofv3 --pmc passes on their LDS / wait counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x, g = bf(M, D), bf(M, D)
W1, W2 = bf(F, D), bf(D, F)
b1 = torch.randn(F, device="cuda") * 0.1
for _ in range(3):
    o.gemm(x, W1, bias=b1)                      # gemm256p_kernel<false, 0>: B k-contiguous (pinned K-tile)
    o.gemm(g, W2, trans_b=True)                 # gemm256p_kernel<true, 0>: B k-strided (transposing LDS reads)
    o.wgrad(g, x)                               # gemm256_kernel<true, true>: both operands k-strided, split-K
torch.cuda.synchronize()
