import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from tools.microbench import timeit
    M = 50176
    for name, n, k in (("qkv", 2304, 768), ("proj", 768, 768), ("fc2", 768, 3072), ("fc1", 3072, 768)):
        a = torch.randn(M, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
        dy = torch.randn(M, n, device="cuda").bfloat16()
        t = timeit(lambda: o.gemm(a, w), iters=20)
        t2 = timeit(lambda: o.gemm(dy, w, trans_b=True), iters=20)
        t3 = timeit(lambda: o.wgrad(dy, a), iters=20)
        fl = 2.0 * M * n * k / 1e9
        print(f"  {name:5s} fwd {t*1e3:7.1f} us {fl/t:7.1f} TF | dgrad {t2*1e3:7.1f} us {fl/t2:7.1f} TF | wgrad {t3*1e3:7.1f} us {fl/t3:7.1f} TF")
else:
    for env in sys.argv[1:]:
        e = dict(os.environ)
        for kv in env.split(","):
            if "=" in kv:
                k, v = kv.split("="); e[k] = v
        print(f"== {env}")
        sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "child"], env=e)
