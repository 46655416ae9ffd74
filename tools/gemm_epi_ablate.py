"""Where does a fused epilogue's time go?  Times the two heavy epilogues of the encoder block (fc1 + bias + GELU + saved pre-activation;
dfc2 + dGELU + column sums) on the one-tile-per-workgroup 256 x 256 kernel with parts of the epilogue compiled out, using a
-DDEVIAS_GEMM_DEBUG build of the library (built here into tools/exp/libdevias_amd_dbg.so).  gemm_debug bits: 2 = no epilogue,
64 = no C stores, 128 = no pre-activation stores, 256 = no GELU / dGELU polynomial.
Usage: python tools/gemm_epi_ablate.py [--rebuild]   (builds the debug library if it is missing, then runs the child)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "tools", "exp", "libdevias_amd_dbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from devias_amd import ops as o
    from devias_amd._lib import ACT_GELU, ACT_DGELU
    from tools.microbench import timeit
    M, D, F = 50176, 768, 3072
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    u, g = bf(M, D), bf(M, D)
    W1, W2 = bf(F, D), bf(D, F)
    b1 = torch.randn(F, device="cuda") * 0.1
    hpre = bf(M, F)
    db1 = torch.zeros(F, device="cuda")
    o.set_option("gemm_persistent", 0);
    for bits in (0, 256, 128, 64, 64 | 128, 64 | 128 | 256, 2):
        o.set_option("gemm_debug", bits)
        t1 = timeit(lambda: o.gemm(u, W1, bias=b1, act=ACT_GELU, aux_out=hpre), iters=20)
        t2 = timeit(lambda: o.gemm(g, W2, trans_b=True, act=ACT_DGELU, aux_in=hpre, colsum=db1), iters=20)
        t3 = timeit(lambda: o.gemm(u, W1, bias=b1), iters=20)
        print(f"  debug={bits:3d}  fc1+gelu+aux {t1*1e3:7.1f} us   dfc2+dgelu+colsum {t2*1e3:7.1f} us   fc1 bias only {t3*1e3:7.1f} us")
else:
    from devias_amd import build as b
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
    if not os.path.exists(DBG) or "--rebuild" in sys.argv:          # (built on the CPU container; the .so travels to the GPU box)
        objs = []
        for s in b.SOURCES:
            obj = os.path.join(ROOT, "tools", "exp", s.replace(".hip", ".dbg.o"))
            subprocess.run([b.HIPCC] + b.FLAGS + ["-DDEVIAS_GEMM_DEBUG", "-c", os.path.join(b.CSRC, s), "-o", obj], check=True)
            objs.append(obj)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DBG] + objs, check=True)
    e = dict(os.environ, DEVIAS_LIB_PATH=DBG)
    subprocess.run([sys.executable, __file__, "child"], env=e)
