import sys; sys.path.insert(0, "/root/repo")
import torch
from devias_amd import ops as o
from tools.microbench import timeit
M, D = 50176, 768
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
u, Wqkv, g3 = bf(M, D), bf(3 * D, D), bf(M, 3 * D)
bq = torch.randn(3 * D, device="cuda") * 0.1
print("qkv fwd  %7.1f us" % (timeit(lambda: o.gemm(u, Wqkv, bias=bq), iters=20) * 1e3))
print("dqkv     %7.1f us" % (timeit(lambda: o.gemm(g3, Wqkv, trans_b=True), iters=20) * 1e3))
