import sys; sys.path.insert(0, "/root/repo")
import torch
from devias_amd import ops as o
from tools.microbench import timeit
M = 50176
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
for name, n, k in (("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("proj", 768, 768)):
    a, w = bf(M, k), bf(n, k)
    b = torch.randn(n, device="cuda")
    c = o.gemm(a, w, bias=b)
    ref = (a[:512].float() @ w.float().t() + b)
    err = float((c[:512].float() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: o.gemm(a, w, bias=b), iters=20)
    print(f"{name:5s} {t*1e3:7.1f} us {2.0*M*n*k/t/1e9:7.1f} TF  err {err:.2e}")
