#!/usr/bin/env python3
"""prologue / slice loop / epilogue split of the one-wave-per-SIMD dK / dV kernel (library built with -DDKDV_STAMP): shader cycles per workgroup"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops, _lib
B, N, H = 32, 1568, 12
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
d_o = torch.randn(B * N, H * 64, device="cuda").to(torch.bfloat16)
o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
for _ in range(3):
    ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125)
torch.cuda.synchronize()
n = 2304
buf = (ctypes.c_uint64 * (8 * n))()
_lib.check(_lib.load().devias_debug_dkdv_stamps(ctypes.cast(buf, ctypes.c_void_p), n), "stamps")
t = torch.tensor(list(buf), dtype=torch.int64).view(n, 8)
pro, loop, epi = (t[:, 1] - t[:, 0]).double(), (t[:, 2] - t[:, 1]).double(), (t[:, 3] - t[:, 2]).double()
t0 = t[:, 0].min()
print(f"per workgroup, cycles (median / p10 / p90): prologue {pro.median():.0f} / {pro.quantile(0.1):.0f} / {pro.quantile(0.9):.0f}; "
      f"slice loop {loop.median():.0f} / {loop.quantile(0.1):.0f} / {loop.quantile(0.9):.0f} = {loop.median() / 49:.0f} per slice; "
      f"epilogue {epi.median():.0f} / {epi.quantile(0.1):.0f} / {epi.quantile(0.9):.0f}")
d = lambda a, b: (t[:, a] - t[:, b]).double().median().item()
print(f"prologue parts (median cycles): entry -> prefill issued {d(4, 0):.0f}; -> K/V fragments + accumulators in AGPRs {d(5, 4):.0f}; -> slice 0 landed + barrier {d(6, 5):.0f}; "
      f"-> first S/dP done, loop entry {d(1, 6):.0f}.  epilogue: drain + tile to LDS {d(7, 2):.0f}; stores {d(3, 7):.0f}")
