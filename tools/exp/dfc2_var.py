import sys; sys.path.insert(0, "/root/repo")
import torch
from devias_amd import ops as o
from devias_amd._lib import ACT_GELU, ACT_DGELU
from tools.microbench import timeit
M, D, F = 50176, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
g, W2, hpre, x = bf(M, D), bf(D, F), bf(M, F), bf(M, F)
db1 = torch.zeros(F, device="cuda")
for name, fn in (("plain", lambda: o.gemm(g, W2, trans_b=True)),
                 ("dgelu", lambda: o.gemm(g, W2, trans_b=True, act=ACT_DGELU, aux_in=hpre)),
                 ("colsum", lambda: o.gemm(g, W2, trans_b=True, colsum=db1)),
                 ("res", lambda: o.gemm(g, W2, trans_b=True, res=x)),
                 ("dgelu+colsum", lambda: o.gemm(g, W2, trans_b=True, act=ACT_DGELU, aux_in=hpre, colsum=db1))):
    t = timeit(fn, iters=20)
    print(f"{name:14s} {t*1e3:7.1f} us")
