"""backward attention at the bench shape: 20 launches of devias_mhsa_bwd (dQ kernel + dK/dV kernel) timed by HIP events, per attn_cfg in argv (default 0);
per-kernel times: run under rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
B, N, H = 32, 1568, 12
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 0.5).bfloat16()
d_o = (torch.randn(B * N, H * 64, device="cuda") * 0.5).bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
res = {}
for cfg in [int(a) for a in sys.argv[1:]] or [0]:
    o.set_option("attn_cfg", cfg)
    for _ in range(3): dqkv = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dqkv = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    print(f"attn_cfg {cfg}: mhsa_bwd {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (dQ + dK/dV)")
    if cfg in res: assert torch.equal(res[cfg], dqkv)
    res[cfg] = dqkv
ks = list(res)
for k in ks[1:]: print(f"attn_cfg {k} bitwise equal to attn_cfg {ks[0]}:", torch.equal(res[k], res[ks[0]]))
