"""the four weight-gradient GEMMs of an encoder block at the bench shape (M = 50176 rows reduced): HIP-event time per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
M = 50176
for name, No, Ki in (("wqkv", 2304, 768), ("wproj", 768, 768), ("wfc1", 3072, 768), ("wfc2", 768, 3072)):
    g = (torch.randn(M, No, device="cuda") * 0.1).bfloat16(); x = (torch.randn(M, Ki, device="cuda") * 0.5).bfloat16()
    out = torch.empty(No, Ki, device="cuda")
    for _ in range(3): o.wgrad(g, x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o.wgrad(g, x, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: {us:.1f} us per launch (+ reduce) = {2.0 * M * No * Ki / us / 1e6:.0f} TFLOP/s, checksum {float(out.abs().sum()):.6e}")
