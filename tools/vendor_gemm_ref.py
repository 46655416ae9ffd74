"""Calibration only (not product, not used by anything): what the vendor GEMM library behind torch.matmul (hipBLASLt / rocBLAS) reaches on the encoder block's
GEMM shapes on this box, next to this library's kernels with the plain epilogues (bias only / none).  M = 50176, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x, xf = bf(M, D), bf(M, F)
g3 = bf(M, 3 * D)
Wqkv, Wp, W1, W2 = bf(3 * D, D), bf(D, D), bf(F, D), bf(D, F)
b1 = (torch.randn(F, device="cuda") * 0.1)
b1h = b1.bfloat16()
rows = [
    ("qkv  fwd  [M,768]x[2304,768]^T", 2 * M * 3 * D * D, lambda: o.gemm(x, Wqkv), lambda: torch.nn.functional.linear(x, Wqkv)),
    ("proj fwd  [M,768]x[768,768]^T", 2 * M * D * D, lambda: o.gemm(x, Wp), lambda: torch.nn.functional.linear(x, Wp)),
    ("fc1  fwd  + bias", 2 * M * F * D, lambda: o.gemm(x, W1, bias=b1), lambda: torch.nn.functional.linear(x, W1, b1h)),
    ("fc2  fwd  [M,3072]x[768,3072]^T", 2 * M * F * D, lambda: o.gemm(xf, W2), lambda: torch.nn.functional.linear(xf, W2)),
    ("dfc2 dgrad [M,768]x[768,3072]", 2 * M * F * D, lambda: o.gemm(x, W2, trans_b=True), lambda: x @ W2),
    ("dfc1 dgrad [M,3072]x[3072,768]", 2 * M * F * D, lambda: o.gemm(xf, W1, trans_b=True), lambda: xf @ W1),
    ("dqkv dgrad [M,2304]x[2304,768]", 2 * M * 3 * D * D, lambda: o.gemm(g3, Wqkv, trans_b=True), lambda: g3 @ Wqkv),
    ("wfc1 wgrad [3072,M]x[M,768] (fp32 out here, bf16 out there)", 2 * M * F * D, lambda: o.wgrad(xf, x), lambda: xf.t() @ x),
    ("wproj wgrad [768,M]x[M,768]", 2 * M * D * D, lambda: o.wgrad(x, x), lambda: x.t() @ x),
]
print(f"{'shape':62s} {'this library':>22s} {'torch.matmul (vendor)':>24s}")
for name, fl, mine, ref in rows:
    t1 = timeit(mine, iters=20); t2 = timeit(ref, iters=20)
    print(f"{name:62s} {t1*1e3:8.1f} us {fl/t1/1e9:7.0f} TF   {t2*1e3:8.1f} us {fl/t2/1e9:7.0f} TF")
