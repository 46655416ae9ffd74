"""Attention kernels at N = 1536 (12 full 128-row blocks per head), 1568 (the step's: 12.25 blocks -> 13, 24.5 tiles -> 25), 1664 (13 full blocks): what the
mostly-idle last block of a head costs.  B = 32, H = 12, bf16; separate launches timed with rocprof-free HIP events over the pair (dQ + dK/dV) and the forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
B, H = 32, 12
for rep in range(2):
    for N in (1536, 1568, 1600, 1664):
        qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
        out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
        do = torch.randn_like(out)
        t = timeit(lambda: o.mhsa_fwd(qkv, B, N, H, 0.125), iters=40, warmup=20)
        t2 = timeit(lambda: o.mhsa_bwd(qkv, out, do, lse, B, N, H, 0.125), iters=20, warmup=10)
        print(f"N={N}: fwd {t*1e3:7.1f} us ({t*1e3/(N/1536)**2:7.1f} per 1536^2-equivalent)   bwd {t2*1e3:7.1f} us ({t2*1e3/(N/1536)**2:7.1f})", flush=True)
