"""A/B of the attention backward paths in ONE process (interleaved rounds): single pass (attn_bwd = 1) vs dQ + dK/dV kernels (attn_bwd = 0),
B = 32, H = 12, N = 1568 bf16.  Usage: python tools/attn_bwd_ab.py [N] [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1568
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.0).bfloat16()
d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
fl5 = 5 * 2.0 * B * H * N * N * 64
DBG = [int(x) for x in os.environ.get("DBGS", "0").split(",")]
modes = [(1, d) for d in DBG] + [(0, 0)]
times = {m: [] for m in modes}
for rnd in range(5):
    for mode in modes:
        o.set_option("attn_bwd", mode[0]); o.set_option("attn_dbg", mode[1])
        for _ in range(2):
            o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e1.record()
        torch.cuda.synchronize()
        times[mode].append(e0.elapsed_time(e1) / 10)
o.set_option("attn_dbg", 0)
for mode in modes:
    t = sorted(times[mode])
    print(f"attn_bwd,dbg={mode}: median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us   {fl5 / t[len(t)//2] / 1e9:.0f} TF/s algorithmic (5 products)")
print("hand-off timeouts:", o.mhsa_bwd_handoff_timeouts())
