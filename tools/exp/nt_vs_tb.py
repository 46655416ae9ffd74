"""dgrad as NT on a transposed weight copy against dgrad with the weight read k-strided (trans_b): the block's four dgrad shapes, alternating launches, same box.
usage: python tools/exp/nt_vs_tb.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
shapes = [("dfc2 (plain)", D, F), ("dfc1", F, D), ("dproj", D, D), ("dqkv", 3 * D, D)]      # (name, reduction = out features, N = in features)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, n=10):
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, K, N in shapes:
    dy = bf(M, K); W = (torch.randn(K, N, device="cuda") * 0.05).bfloat16(); Wt = W.t().contiguous()
    a = o.gemm(dy, W, trans_b=True); b = o.gemm(dy, Wt)
    same = torch.equal(a, b)
    for _ in range(3): o.gemm(dy, W, trans_b=True); o.gemm(dy, Wt)
    tb, nt = [], []
    for r in range(6):
        tb.append(t(lambda: o.gemm(dy, W, trans_b=True))); nt.append(t(lambda: o.gemm(dy, Wt)))
    print(f"{name:14s} [M,{K}]x[{K},{N}]: k-strided W {min(tb):7.1f} us (med {sorted(tb)[3]:.1f})   transposed copy (NT) {min(nt):7.1f} us (med {sorted(nt)[3]:.1f})   bitwise equal: {same}")
