"""A/B of the dK/dV kernels in ONE process (interleaved rounds): generation 1 (attn_dkdv = 0) vs generation 2 variants, bf16, with a correctness check
of each variant against the fp32 reference on a small shape and against generation 1 on the measured shape.
Usage: python tools/attn_dkdv_ab.py [cfgs comma separated, default 0,42,22] [N] [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o

cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,42,22").split(",")]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1568
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H = int(sys.argv[4]) if len(sys.argv) > 4 else 12


def ref(qkv, d_o, B, N, H):
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    out = (((q * 0.125) @ k.transpose(-1, -2)).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    out.backward(d_o.float())
    return x.grad.reshape(B, N, 3, H, 64)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


for (b_, n_, h_) in ((2, 200, 2), (1, 97, 1), (2, 1568, 2), (1, 33, 3)):
    qkv = torch.randn(b_ * n_, 3 * h_ * 64, device="cuda").bfloat16()
    d_o = torch.randn(b_ * n_, h_ * 64, device="cuda").bfloat16()
    out, lse = o.mhsa_fwd(qkv, b_, n_, h_, 0.125)
    g = ref(qkv, d_o, b_, n_, h_)
    for c in cfgs:
        o.set_option("attn_dkdv", c)
        d = o.mhsa_bwd(qkv, out, d_o, lse, b_, n_, h_, 0.125).float().reshape(b_, n_, 3, h_, 64)
        print(f"B={b_} N={n_} H={h_} dkdv={c}: dq {rel(d[:, :, 0], g[:, :, 0]):.2e} dk {rel(d[:, :, 1], g[:, :, 1]):.2e} dv {rel(d[:, :, 2], g[:, :, 2]):.2e}")

qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
o.set_option("attn_dkdv", 0)
base = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125).float()
times = {c: [] for c in cfgs}
for rnd in range(5):
    for c in cfgs:
        o.set_option("attn_dkdv", c)
        for _ in range(2):
            o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            r = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125)
        e1.record()
        torch.cuda.synchronize()
        times[c].append(e0.elapsed_time(e1) / 10)
for c in cfgs:
    o.set_option("attn_dkdv", c)
    r = o.mhsa_bwd(qkv, out, d_o, lse, B, N, H, 0.125).float()
    t = sorted(times[c])
    print(f"attn_dkdv={c}: dQ + dK/dV median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us; vs generation 1: rel {rel(r, base):.2e}; finite {bool(torch.isfinite(r).all())}")
o.set_option("attn_dkdv", 0)
