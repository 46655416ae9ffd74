"""one attention forward + backward at the measured shape with a chosen dK/dV kernel: target of rocprofv3 PMC passes.  python tools/attn_one.py <attn_dkdv> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
c = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, N, H = 32, 1568, 12
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
o.set_option("attn_dkdv", c)
for _ in range(reps):
    out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
    o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
torch.cuda.synchronize()
