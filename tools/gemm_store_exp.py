import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from tools.microbench import timeit
    N, K = 2304, 64
    for M in (256 * 7, 256 * 14, 256 * 28, 256 * 56, 256 * 196):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: o.gemm(a, w, out=out), iters=50, warmup=5)
        print(f"  M={M:6d} tiles={M//256*9:5d}  {t*1e3:7.2f} us   C bytes {M*N*2/1e6:7.1f} MB -> {M*N*2/t/1e9:6.2f} TB/s")
else:
    for env in sys.argv[1:]:
        e = dict(os.environ)
        for kv in env.split(","):
            if "=" in kv:
                k, v = kv.split("="); e[k] = v
        print(f"== {env}"); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "child"], env=e)
