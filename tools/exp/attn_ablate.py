"""timing ablations of the generation-2 dK/dV kernel (results are wrong by construction): attn_dkdv = 100 * ablation bits + 42"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
B, N, H = 32, 1568, 12
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
names = {0: "generation 1", 42: "generation 2", 142: "gen2 - DMA/waits/barriers", 242: "gen2 - softmax arithmetic", 442: "gen2 - S/dP MFMAs", 842: "gen2 - dV/dK MFMAs",
         1242: "gen2 - all MFMAs", 342: "gen2 - DMA/barriers - softmax", 1442: "gen2 - softmax - all MFMAs"}
res = {}
for rnd in range(3):
    for c in names:
        o.set_option("attn_dkdv", c)
        for _ in range(2):
            o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(c, []).append(e0.elapsed_time(e1) / 10)
base = None
for c, n in names.items():
    t = sorted(res[c])[1] * 1e3
    print(f"{n:40s} dQ + dK/dV {t:7.1f} us")
