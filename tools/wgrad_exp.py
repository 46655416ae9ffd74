import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from tools.microbench import timeit
    M = 50176
    for name, n, k in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        a = torch.randn(M, k, device="cuda").bfloat16(); dy = torch.randn(M, n, device="cuda").bfloat16()
        fl = 2.0 * M * n * k / 1e9
        row = []
        for sk in (0, 4, 7, 9, 14, 28):
            if sk == 0:
                t = timeit(lambda: o.wgrad(dy, a), iters=20)
            else:
                t = timeit(lambda: o.gemm(dy, a, trans_a=True, trans_b=True, out_f32=True, split_k=sk), iters=20)
            row.append(f"sk={sk}:{t*1e3:6.1f}us")
        print(f"  {name:5s} " + "  ".join(row))
else:
    for env in sys.argv[1:]:
        e = dict(os.environ)
        for kv in env.split(","):
            if "=" in kv:
                k, v = kv.split("="); e[k] = v
        print(f"== {env}"); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "child"], env=e)
