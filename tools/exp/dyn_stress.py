"""Stress of the dynamic tile queue of gemm256p_kernel: concurrent kernels on side streams (late / displaced workgroups), every output compared
bitwise with the one-tile-per-workgroup kernel.  usage: dyn_stress.py [child <mode> <debug>]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o, _lib
    from devias_amd._lib import ACT_GELU, ACT_DGELU
    mode, dbg = sys.argv[2], int(sys.argv[3])
    torch.manual_seed(0)
    M, D = 256 * 196, 768
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    A, W, W1, W2t = bf(M, D), bf(D, D) * 0.1, bf(4 * D, D) * 0.1, bf(D, 4 * D) * 0.1
    b, b1 = torch.randn(D, device="cuda"), torch.randn(4 * D, device="cuda")
    res, pre = bf(M, D), bf(M, 4 * D)
    def run():
        return (o.gemm(A, W, bias=b, res=res), o.gemm(A, W1, bias=b1, act=ACT_GELU), o.gemm(A, W2t, trans_b=True, act=ACT_DGELU, aux_in=pre),
                o.gemm(A, W.t().contiguous(), trans_b=True))
    o.set_option("gemm_persistent", 0)
    ref = run()
    o.set_option("gemm_persistent", 1); o.set_option("gemm_dynamic", 1); o.set_option("gemm_debug", dbg)
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    lib = _lib.load()
    bad = 0
    for it in range(40):
        if mode in ("junk", "both"):
            with torch.cuda.stream(side):
                junk.add_(1.0)
        if mode in ("hog", "both") and it % 4 == 0:
            _lib.check(lib.devias_debug_cu_hog(24 + it % 17, 3000, side2.cuda_stream), "hog")
        got = run()
        torch.cuda.synchronize()
        for k, (x, y) in enumerate(zip(got, ref)):
            if not torch.equal(x, y):
                d = (x.float() - y.float()).abs()
                rows = (d.amax(1) > 0).nonzero().flatten()
                cols = (d.amax(0) > 0).nonzero().flatten()
                print(f"  it {it} gemm {k}: {int((d > 0).sum())} elements differ, rows {int(rows.min())}..{int(rows.max())} ({rows.numel()}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()}), max {float(d.max()):.3g}", flush=True)
                bad += 1
    print(f"mode {mode} debug {dbg}: {bad} mismatching outputs of {40 * 4}; dynamic launches {o.counters()['gemm256d']}", flush=True)
else:
    for mode in ("none", "junk", "hog", "both"):
        for dbg in (0, 512):
            r = subprocess.run([sys.executable, __file__, "child", mode, str(dbg)], capture_output=True, text=True)
            out = "\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l)
            print(f"== {mode} debug={dbg} rc={r.returncode}\n" + "\n".join(out.splitlines()[-12:]), flush=True)
