"""What holds the weight-gradient kernel's K loop at 2.0 us per K-tile?  The fc1 weight gradient dW[3072, 768] = dY^T X (reduction over M) three ways, same split-K (one round of
workgroups), alternating launches: (a) as the step runs it, both operands k-strided (transposing LDS reads, compiler-scheduled K-tile); (b) the SAME product on pre-transposed
operands dY^T [3072, M], X^T [768, M] -- both k-contiguous: the pinned K-tile of the forward GEMMs, same HBM streams; (c) both at M / 8 (operands resident in the Infinity Cache).
If (b) runs at the forward kernels' 1.6 us per K-tile the loss is the k-strided schedule; if (b) is as slow as (a) and (c) is fast, it is the HBM stream / the 2-stage ring."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
D, F = 768, 3072
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, n=10):
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (50176, 6272):
    dy = (torch.randn(M, F, device="cuda") * 0.5).bfloat16(); x = (torch.randn(M, D, device="cuda") * 0.5).bfloat16()
    dyT, xT = dy.t().contiguous(), x.t().contiguous()
    sk = o.auto_split_k(F, D, M, bk=64)
    ws = torch.empty(F, D, device="cuda")
    f_tn = lambda: o.gemm(dy, x, trans_a=True, trans_b=True, out=ws, out_f32=True, split_k=sk)
    f_nt = lambda: o.gemm(dyT, xT, out=ws, out_f32=True, split_k=sk)
    a = f_tn().clone(); b = f_nt().clone()
    o.counters(reset=True); f_nt(); cnt = o.counters()
    for _ in range(3): f_tn(); f_nt()
    tn, nt = [], []
    for r in range(6):
        tn.append(t(f_tn)); nt.append(t(f_nt))
    ktiles = M / sk / 64
    print(f"M = {M}: split {sk} ({ktiles:.0f} K-tiles per workgroup); k-strided operands {min(tn):7.1f} us = {min(tn) / ktiles:.2f} us per K-tile;  pre-transposed (both k-contiguous) {min(nt):7.1f} us = "
          f"{min(nt) / ktiles:.2f} us per K-tile   (incl. the split-K reduce ~12 us; equal: {torch.equal(a, b)}; kernels {dict((k, v) for k, v in cnt.items() if v)})")
