#!/usr/bin/env python3
"""attention forward at the step's shape: time per call (HIP events over 20 calls, 5 repetitions) and a checksum of the output, for comparing library builds
(DEVIAS_LIB_PATH=...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops
B, N, H = 32, 1568, 12
torch.manual_seed(1)
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
v = []
for rep in range(5):
    for _ in range(3):
        o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    v.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"{os.environ.get('DEVIAS_LIB_PATH', 'default library')}: forward {sorted(v)[2]:.1f} us (min {min(v):.1f}); checksum {o.float().sum().item():.6f} {lse.sum().item():.6f}", flush=True)
