"""devias_mhsa_bwd_bias at the bench shape: bias gradients from the kernels' accumulators (attn_bias_fused = 1) against two column-sum passes (0), alternating, HIP events"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
B, N, H = 32, 1568, 12
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 0.5).bfloat16()
d_o = (torch.randn(B * N, H * 64, device="cuda") * 0.5).bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
dbq = torch.empty(H * 64, device="cuda"); dbv = torch.empty(H * 64, device="cuda")
for mode in ((0, 1, 1, 0, 0, 1, -1, -1) if "DEVIAS_ATTN_BIAS_FUSED" not in os.environ else (9, 9, 9)):
    if 0 <= mode < 9: o.set_option("attn_bias_fused", mode)
    f = (lambda: o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))) if mode >= 0 else (lambda: o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"attn_bias_fused {mode if mode >= 0 else 'n/a (plain backward, no bias gradients)'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
