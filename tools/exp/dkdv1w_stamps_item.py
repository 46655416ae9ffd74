#!/usr/bin/env python3
"""the persistent one-wave-per-SIMD dK / dV kernel, second item of every workgroup (library built with -DDKDV_STAMP -DDKDV_STAMP_ITEM=1): what the switch between two
256-key blocks costs, in shader cycles"""
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
n = 256
buf = (ctypes.c_uint64 * (8 * n))()
_lib.check(_lib.load().devias_debug_dkdv_stamps(ctypes.cast(buf, ctypes.c_void_p), n), "stamps")
t = torch.tensor(list(buf), dtype=torch.int64).view(n, 8)
d = lambda a, b: (t[:, a] - t[:, b]).double().median().item()
print(f"median cycles per workgroup: item 0 loop exit -> its stores issued {d(5, 4):.0f}; -> K/V in AGPRs, accumulators zeroed {d(0, 5):.0f}; -> slice 0 ready (wait + barrier) {d(6, 0):.0f}; "
      f"-> loop entry {d(1, 6):.0f}  [switch total {d(1, 4):.0f}];  item 1: slice loop {d(2, 1):.0f} = {d(2, 1) / 49:.0f} per slice; drain + tile to LDS {d(7, 2):.0f}; stores {d(3, 7):.0f}")
