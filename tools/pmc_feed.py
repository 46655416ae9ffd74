#!/usr/bin/env python3
"""A few launches of the K = 768 GEMMs of the encoder block (proj with bias + residual, its dgrad, qkv, fc1 with bias only) for rocprofv3 --pmc passes on the CU's
L2 -> LDS feed path (TCP -> TCC requests and latency, TCC hit rate, TA busy, LDS instruction counters): VERDICT r3 item 6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x, g, res = bf(M, D), bf(M, D), bf(M, D)
Wp, Wqkv, W1 = bf(D, D), bf(3 * D, D), bf(F, D)
b = torch.randn(D, device="cuda") * 0.1
b3, b1 = torch.randn(3 * D, device="cuda") * 0.1, torch.randn(F, device="cuda") * 0.1
for _ in range(3):
    o.gemm(x, Wp, bias=b, res=res)              # gemm256p_kernel<false, 1, false>: proj (588 tiles, 12 K-tiles)
    o.gemm(g, Wp, trans_b=True)                 # gemm256p_kernel<true, 0, false>: dproj
    o.gemm(x, Wqkv, bias=b3)                    # gemm256p_kernel<false, 0, false>: qkv (1764 tiles)      } same instantiation: the summary's
    o.gemm(x, W1, bias=b1)                      #                                   fc1, bias only (2352)  } median is over both
torch.cuda.synchronize()
