"""phase timeline of the 64-keys-per-wave dK/dV kernel (library built with -DDEVIAS_ATTN_STAMPS): shader cycles per tile spent waiting for the
LDS-DMA, at the workgroup barrier, in S/dP + softmax, in the dV MFMAs and in the dK MFMAs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
B, N, H = 32, 1568, 12
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
o.set_option("attn_dkdv", nw)
for _ in range(3):
    dqkv = torch.empty_like(qkv)
    delta = torch.zeros((B, H, N), dtype=torch.float32, device="cuda")
    from devias_amd import _lib
    _lib.check(_lib.load().devias_mhsa_bwd(qkv.data_ptr(), out.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, N, H, 0.125, 1, None,
                                           torch.cuda.current_stream().cuda_stream), "bwd")
    torch.cuda.synchronize()
d = delta.view(-1).view(torch.int64)[: 16 * nw * 8].view(-1, 8).cpu()
rows = [r for r in d.tolist() if r[5] == 49]
print(f"{len(rows)} waves sampled (nw = {nw}); cycles per tile: wait-DMA, barrier, S/dP+softmax, dV MFMAs, dK MFMAs, total")
for r in rows[:12]:
    per = [x / 49 for x in r[:5]]
    print("  " + "  ".join(f"{x:7.0f}" for x in per) + f"  | {sum(per):7.0f}")
