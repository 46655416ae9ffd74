import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
M, D = 50176, 768
x = torch.randn(M, D, device="cuda").bfloat16(); dy = torch.randn(M, D, device="cuda").bfloat16(); dr = torch.randn(M, D, device="cuda").bfloat16()
g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
y, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6)
cs = torch.empty(D, device="cuda")
for _ in range(5): o.layernorm_bwd(dy, x, g, mean, rstd, dres=dr, dx_colsum=cs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): o.layernorm_bwd(dy, x, g, mean, rstd, dres=dr, dx_colsum=cs)
e1.record(); torch.cuda.synchronize()
print(f"layernorm_bwd (+ param reduce) M={M} D={D}: {e0.elapsed_time(e1)/50*1e3:.1f} us")
for _ in range(5): o.layernorm_fwd(x, g, b, 1e-6)
e0.record()
for _ in range(50): o.layernorm_fwd(x, g, b, 1e-6)
e1.record(); torch.cuda.synchronize()
print(f"layernorm_fwd M={M} D={D}: {e0.elapsed_time(e1)/50*1e3:.1f} us")
