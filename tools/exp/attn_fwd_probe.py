"""forward attention at the bench shape, attn_cfg from argv (0 = default, 8 = the V image with the old swizzle): timing by HIP events + output checksum"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
B, N, H = 32, 1568, 12
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 0.5).bfloat16()
outs = {}
for cfg in [int(a) for a in sys.argv[1:]] or [0, 8, 8, 0]:
    o.set_option("attn_cfg", cfg)
    for _ in range(3): out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    print(f"attn_cfg {cfg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
    outs[cfg] = out
ks = list(outs)
if len(ks) > 1: print("outputs bitwise equal:", torch.equal(outs[ks[0]], outs[ks[1]]))
