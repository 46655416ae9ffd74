#!/usr/bin/env python3
"""A few launches of each hot kernel at the BASELINE shapes (B=32, ViT-B 16x224^2), for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o

dev, bf = "cuda", torch.bfloat16
B, N, D, H = 32, 1568, 768, 12
M = B * N
x = torch.randn(M, D, device=dev).to(bf)
w1 = (torch.randn(4 * D, D, device=dev) * 0.02).to(bf)
w1t = w1.t().contiguous()
dy = torch.randn(M, 4 * D, device=dev).to(bf)
bias = torch.zeros(4 * D, device=dev)
pre = torch.empty(M, 4 * D, device=dev, dtype=bf)
qkv = torch.randn(M, 3 * D, device=dev).to(bf)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
qp = (torch.randn(B * 2, 4 * D, device=dev) * 0.5).to(bf)
dres = torch.randn(M, D, device=dev).to(bf)
for it in range(3):
    o.gemm(x, w1, bias=bias, act=o.ACT_GELU, aux_out=pre)            # fc1 forward (gemm256p_kernel<false, 0, false, 25>: persistent, B k-contiguous, bias + GELU + second output)
    o.gemm(dy, w1t)                                                    # fc1 dgrad on the transposed weight copy, as the step runs it since round 6 (gemm256p_kernel<false, 0, false, 0>: 588 tiles, K = 3072, tail tiles in thirds)
    o.wgrad(dy, x)                                                     # fc1 wgrad   (gemm256_kernel<true, true> + split-K reduce)
    out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
    o.mhsa_bwd(qkv, out, out, lse, B, N, H, 0.125)
    y, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6)
    o.layernorm_bwd(y, x, g, mean, rstd, dres=dres)                  # three DISTINCT inputs, as in the step (until round 6's second session dres aliased x: 231 MB instead of 308)
    A, r, z = o.slotf_fwd(qp, x, B, 2, N, 4, D, 512 ** -0.5)           # folded slot attention, one layer
    o.slotf_bwd(x, A, r, z, qp, None, B, 2, N, 4, D, 512 ** -0.5)
torch.cuda.synchronize()
