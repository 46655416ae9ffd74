import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from tools.microbench import timeit
    M = 50176
    for n in (2304, 768):
        row = []
        for k in (64, 128, 256, 512, 768, 1536, 3072):
            a = torch.randn(M, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
            t = timeit(lambda: o.gemm(a, w), iters=20)
            row.append(f"K={k}:{t*1e3:6.1f}us")
        print(f"  N={n}: " + "  ".join(row))
else:
    for env in sys.argv[1:]:
        e = dict(os.environ)
        for kv in env.split(","):
            if "=" in kv:
                k, v = kv.split("="); e[k] = v
        print(f"== {env}"); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "child"], env=e)
