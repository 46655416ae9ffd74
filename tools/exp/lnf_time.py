import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit
M, D = 50176, 768
x = torch.randn(M, D, device="cuda").bfloat16(); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
y, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6)
ref = torch.nn.functional.layer_norm(x.float(), (D,), g, b, 1e-6)
err = float((y.float() - ref).abs().max())
ts = [timeit(lambda: o.layernorm_fwd(x, g, b, 1e-6), iters=50) * 1e6 for _ in range(3)]
xl = torch.randn(M, 1024, device="cuda").bfloat16(); gl = torch.ones(1024, device="cuda"); bl = torch.zeros(1024, device="cuda")
tl = [timeit(lambda: o.layernorm_fwd(xl, gl, bl, 1e-6), iters=50) * 1e6 for _ in range(2)]
print(f"  ln_fwd [50176, 768] bf16: " + " ".join(f"{t:.1f}" for t in ts) + f" us  ({2 * M * D * 2 / min(ts) / 1e6:.2f} TB/s)  max err {err:.2e};  [50176, 1024]: " + " ".join(f"{t:.1f}" for t in tl) + " us")
