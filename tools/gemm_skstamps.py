"""Per-item timeline of the stream-K GEMM kernel (debug build, gemm_debug = 8, stream-K forced on): K-tile time under the desynchronised schedule, the cost of
publishing a head fragment, and the wait + load of the predecessor's partial.  Usage: python tools/gemm_skstamps.py [proj|fc2|qkv|fc1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEVIAS_LIB_PATH", os.path.join(ROOT, "tools", "exp", "libdevias_amd_dbg.so"))
import ctypes, torch, numpy as np
from devias_amd import ops as o, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "proj"
M = 50176
N, K = {"fc1": (3072, 768), "proj": (768, 768), "qkv": (2304, 768), "fc2": (768, 3072)}[which]
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
bias = torch.randn(N, device="cuda") * 0.1
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
skws = o.streamk_workspace(a.device)
o.set_option("gemm_streamk", 3); o.set_option("gemm_persistent", 1); o.set_option("gemm_debug", 8)
def call():
    g = _lib.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K; g.lda, g.ldb, g.ldc = K, K, N
    g.dtype = 1; g.split_k = 1; g.ws = ws.data_ptr(); g.bias = bias.data_ptr()
    g.sk_ws, g.sk_ws_bytes = skws.data_ptr(), skws.numel()
    _lib.check(_lib.load().devias_gemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "gemm")
for _ in range(3): call()
torch.cuda.synchronize()
assert o.counters()["gemm_sk"] >= 3
d = ws.cpu().numpy().reshape(256, 64)
t = (d >> 4) / 100.0; code = d & 15
nk = K // 64
kl, ep, waitp, loadp, span = [], [], [], [], []
for b in range(256):
    ev = [(t[b, i], int(code[b, i])) for i in range(63) if code[b, i] != 0]
    if not ev: continue
    span.append(ev[-1][0] - ev[0][0])
    for i in range(len(ev) - 1):
        (ta, ca), (tb, cb) = ev[i], ev[i + 1]
        if (ca, cb) == (1, 2): kl.append(tb - ta)
        if (ca, cb) == (2, 3): ep.append(tb - ta)
        if (ca, cb) == (3, 5): waitp.append(tb - ta)
        if (ca, cb) == (5, 6): loadp.append(tb - ta)
ent = t[:, 63]; first_kt = t[:, 0]; last = np.array([max(t[b, i] for i in range(63) if code[b, i] != 0) for b in range(256)])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
print(f"{which}: back-to-back launch time {e0.elapsed_time(e1) / 20 * 1e3:.1f} us;  workgroup entry spread {ent.max() - ent.min():.1f} us;  entry -> first K-tile multiplied: med {np.median(first_kt - ent):.2f} max {np.max(first_kt - ent):.2f} us;  "
      f"first entry -> last item finished {last.max() - ent.min():.1f} us")
print(f"{which}: N={N} K={K}  per-workgroup span med {np.median(span):.1f} max {np.max(span):.1f} us")
print(f"  item K loops: med {np.median(kl):.2f} us (whole tiles {nk} K-tiles -> {np.percentile(kl, 75) / nk:.3f} us per K-tile at p75)")
print(f"  epilogue / partial store: med {np.median(ep):.2f}  p90 {np.percentile(ep, 90):.2f} us")
print(f"  predecessor's partial: flag wait + 256 KiB load med {np.median(loadp):.2f}  p90 {np.percentile(loadp, 90):.2f}  max {np.max(loadp):.2f} us")
