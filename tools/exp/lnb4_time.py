import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
M, D = 50176, 1024
bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
x, dy, dres = bf(M, D), bf(M, D), bf(M, D)
g = torch.rand(D, device="cuda") + 0.5; b = torch.zeros(D, device="cuda")
y, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6)
out = o.layernorm_bwd(dy, x, g, mean, rstd, dres=dres)
dx = out[0]
# fp32 reference
xr = x.float().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy.float())
e_dx = float(((dx.float() - (xr.grad + dres.float())).abs().max()) / (xr.grad + dres.float()).abs().max())
e_dg = float((out[1] - gr.grad).abs().max() / gr.grad.abs().max())
again = o.layernorm_bwd(dy, x, g, mean, rstd, dres=dres)
same = all(torch.equal(a, c) for a, c in zip(out, again) if isinstance(a, torch.Tensor))
ts = [timeit(lambda: o.layernorm_bwd(dy, x, g, mean, rstd, dres=dres), iters=50) * 1e3 for _ in range(3)]
print(f"  ln_bwd [50176, 1024] bf16 + residual gradient: " + " ".join(f"{t:.1f}" for t in ts) + f" us ({4 * M * D * 2 / min(ts) / 1e6:.2f} TB/s); dx err {e_dx:.1e}, dgamma err {e_dg:.1e}, bitwise run to run {same}")
