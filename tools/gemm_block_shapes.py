"""Time the ten GEMM calls of one encoder block exactly as modeling_slot.EncoderBlockFn issues them (with their fused epilogues),
M = 50176 (B = 32 clips x 1568 tokens), bf16.  Usage: python tools/gemm_block_shapes.py [ENV=VAL,...]... (one child per set)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from devias_amd._lib import ACT_GELU, ACT_DGELU
    from tools.microbench import timeit
    M, D, F = 50176, 768, 3072
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    x, u, oo, g = bf(M, D), bf(M, D), bf(M, D), bf(M, D)
    Wqkv, Wp, W1, W2 = bf(3 * D, D), bf(D, D), bf(F, D), bf(D, F)
    bq, bp, b1, b2 = (torch.randn(n, device="cuda") * 0.1 for n in (3 * D, D, F, D))
    hpre, hact, g3 = bf(M, F), bf(M, F), bf(M, 3 * D)
    db1 = torch.zeros(F, device="cuda")
    calls = [
        ("qkv   fwd  bias", 2 * M * 3 * D * D, lambda: o.gemm(u, Wqkv, bias=bq)),
        ("proj  fwd  bias+res", 2 * M * D * D, lambda: o.gemm(oo, Wp, bias=bp, res=x)),
        ("fc1   fwd  bias+gelu+aux", 2 * M * F * D, lambda: o.gemm(u, W1, bias=b1, act=ACT_GELU, aux_out=hpre)),
        ("fc1   fwd  bias only", 2 * M * F * D, lambda: o.gemm(u, W1, bias=b1)),
        ("fc2   fwd  bias+res", 2 * M * F * D, lambda: o.gemm(hact, W2, bias=b2, res=x)),
        ("dfc2  dgrad dgelu+colsum", 2 * M * F * D, lambda: o.gemm(g, W2, trans_b=True, act=ACT_DGELU, aux_in=hpre, colsum=db1)),
        ("dfc2  dgrad plain", 2 * M * F * D, lambda: o.gemm(g, W2, trans_b=True)),
        ("dfc1  dgrad", 2 * M * F * D, lambda: o.gemm(hact, W1, trans_b=True)),
        ("dproj dgrad", 2 * M * D * D, lambda: o.gemm(g, Wp, trans_b=True)),
        ("dqkv  dgrad", 2 * M * 3 * D * D, lambda: o.gemm(g3, Wqkv, trans_b=True)),
        ("wqkv  wgrad", 2 * M * 3 * D * D, lambda: o.wgrad(g3, u)),
        ("wproj wgrad", 2 * M * D * D, lambda: o.wgrad(g, oo)),
        ("wfc1  wgrad", 2 * M * F * D, lambda: o.wgrad(hact, u)),
        ("wfc2  wgrad", 2 * M * F * D, lambda: o.wgrad(g, hact)),
    ]
    tot = 0.0
    for name, fl, fn in calls:
        t = timeit(fn, iters=20)
        if "only" not in name and "plain" not in name:
            tot += t
        print(f"  {name:28s} {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF")
    print(f"  block total (10 + 4 wgrad) {tot:.3f} ms")
else:
    for env in sys.argv[1:] or [""]:
        e = dict(os.environ)
        for kv in env.split(","):
            if "=" in kv:
                k, v = kv.split("="); e[k] = v
        print(f"== {env}"); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "child"], env=e)
