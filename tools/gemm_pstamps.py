"""Per-tile timeline of the persistent 256 x 256 GEMM kernel (debug build, gemm_debug = 8): where a tile's time goes between the K loop, the
epilogue, and the first K-iteration of the next tile (which has to wait for the epilogue's stores: vmcnt completes in issue order).
Usage: python tools/gemm_pstamps.py [fc1|fc1g|proj|qkv|fc2|dfc2|dfc1|dproj|dqkv]   (needs tools/exp/libdevias_amd_dbg.so: python tools/gemm_epi_ablate.py --rebuild)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEVIAS_LIB_PATH", os.path.join(ROOT, "tools", "exp", "libdevias_amd_dbg.so"))
import ctypes, torch, numpy as np
from devias_amd import ops as o, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "fc1"
M = 50176
N, K = {"fc1": (3072, 768), "fc1g": (3072, 768), "proj": (768, 768), "qkv": (2304, 768), "fc2": (768, 3072),
        "dfc2": (3072, 768), "dfc1": (768, 3072), "dproj": (768, 768), "dqkv": (768, 2304)}[which]
TB = which.startswith("d")
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(*((K, N) if TB else (N, K)), device="cuda") * 0.02).bfloat16()
pre = torch.randn(M, N, device="cuda").bfloat16() if which == "dfc2" else None
csum = torch.zeros(N, device="cuda")
CS = which in ("dfc2", "dproj")                  # fused column sums: the partials own the head of ws ([M / 128][N] floats), the debug build puts its stamps behind them
bias = torch.randn(N, device="cuda") * 0.1
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
res = torch.randn(M, N, device="cuda").bfloat16()
ws = torch.zeros(256 * 64 + (M // 128 * N // 2 if CS else 0), dtype=torch.int64, device="cuda")
o.set_option("gemm_persistent", 2); o.set_option("gemm_debug", 8)
if len(sys.argv) > 2: o.set_option("gemm_dynamic", int(sys.argv[2]))
def call():
    g = _lib.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K; g.lda, g.ldb, g.ldc = (0 if os.environ.get("PSTAMP_LDA0") else K), (N if TB else K), N      # (PSTAMP_LDA0=1: every A row is row 0 -- the A stream from L2 instead of HBM: what the K loop does without A's memory latency)
    g.trans_b = 1 if TB else 0
    g.dtype = 1; g.split_k = 1; g.ws = ws.data_ptr()
    if not TB: g.bias = bias.data_ptr()
    if which == "dfc2":
        g.act = 4; g.aux_in = pre.data_ptr(); g.ld_aux = N
    if CS:
        g.colsum = csum.data_ptr()
    if which == "fc1g":
        g.act = 1; g.aux_out = aux.data_ptr(); g.ld_aux = N
    if which in ("proj", "fc2"):
        g.res = res.data_ptr(); g.ldr = N
    _lib.check(_lib.load().devias_gemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "gemm")
for _ in range(3): call()
torch.cuda.synchronize()
assert o.counters()["gemm256p"] >= 3
d = ws.cpu().numpy()[(M // 128 * N // 2 if CS else 0):].reshape(256, 64)
t = (d >> 4) / 100.0; code = d & 15
kl, ep, first, steady, clk = [], [], [], [], []
ep_a, ep_b, ep_c = [], [], []          # dynamic queue: K loop done -> epilogue stores issued (2 -> 5), dequeue block + decode (5 -> 6), counted wait (6 -> 3)
t0 = t[:, 0].min()
for b in range(256):
    raw = [(int(d[b, i] >> 4), int(code[b, i])) for i in range(64) if code[b, i] != 0]
    # shader-clock pairs (codes 9 -> 10) against the 100 MHz pairs (1 -> 2) of the same K loop
    r1 = [v for v, c in raw if c == 1]; r2 = [v for v, c in raw if c == 2]; c9 = [v for v, c in raw if c == 9]; c10 = [v for v, c in raw if c == 10]
    for i in range(min(len(r1), len(r2), len(c9), len(c10))):
        if r2[i] > r1[i]: clk.append((c10[i] - c9[i]) / (r2[i] - r1[i]) * 100.0)
    ev = [(t[b, i], code[b, i]) for i in range(63) if code[b, i] not in (0, 9, 10)]
    for i in range(len(ev) - 1):
        (ta, ca), (tb, cb) = ev[i], ev[i + 1]
        if (ca, cb) == (1, 4): first.append(tb - ta)
        if (ca, cb) == (4, 2): steady.append(tb - ta)
        if (ca, cb) == (2, 3): ep.append(tb - ta)
        if (ca, cb) == (2, 5): ep_a.append(tb - ta)
        if (ca, cb) == (5, 6): ep_b.append(tb - ta)
        if (ca, cb) == (6, 3): ep_c.append(tb - ta)
nk = K // 64
ent = t[:, 63]; first_kt = np.array([t[b, 0] for b in range(256)]); last = np.array([max(t[b, i] for i in range(63) if code[b, i] not in (0, 9, 10)) for b in range(256)])
span = (last.max() - t0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
print(f"{which}: back-to-back launch time {e0.elapsed_time(e1) / 20 * 1e3:.1f} us;  workgroup entry spread {ent.max() - ent.min():.1f} us;  entry -> first K-tile multiplied: med {np.median(first_kt - ent):.2f} max {np.max(first_kt - ent):.2f} us;  "
      f"first entry -> last epilogue issued {last.max() - ent.min():.1f} us")
print(f"{which}: N={N} K={K}  kernel span {span:.1f} us, tiles/WG max {int((code == 2).sum(1).max())}")
def dist(name, v):
    v = np.asarray(v)
    if len(v): print(f"  {name}: n {len(v)}  mean {v.mean():.2f}  p10 {np.percentile(v, 10):.2f}  p50 {np.median(v):.2f}  p90 {np.percentile(v, 90):.2f}  p99 {np.percentile(v, 99):.2f}  max {v.max():.2f} us")
dist("first K-iteration", first); dist(f"remaining {nk - 1} K-iterations", steady); dist("epilogue interval", ep)
# per workgroup: first stamp -> last stamp, and how the launch's span divides into K loops / epilogues / the rest (ramp, waiting for the slowest workgroup)
busy = np.array([t[b][(code[b] != 0) & (code[b] < 8)].max() - t[b][(code[b] != 0) & (code[b] < 8)].min() for b in range(256)])
print(f"  per workgroup, first K-tile -> last stamp: mean {busy.mean():.1f}  min {busy.min():.1f}  max {busy.max():.1f} us;  sum over the launch: K loops {(np.sum(first) + np.sum(steady)) / 256:.1f} us per CU, epilogue intervals {np.sum(ep) / 256:.1f} us per CU")
print(f"  first K-iteration of a tile (incl. wait for the previous tile's stores): med {np.median(first):.2f}  p90 {np.percentile(first, 90):.2f} us")
print(f"  remaining {nk - 1} K-iterations: med {np.median(steady):.2f} us  -> {np.median(steady) / max(nk - 1, 1):.3f} us per iteration")
print(f"  shader clock during the K loops: med {np.median(clk):.0f} MHz (p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f})")
if ep: print(f"  epilogue (K loop done -> stores issued, next tile's first K-tile landed): med {np.median(ep):.2f}  p90 {np.percentile(ep, 90):.2f} us")
if ep_a: print(f"  dynamic queue: epilogue stores issued {np.median(ep_a):.2f} us, dequeue + publish + read + decode {np.median(ep_b):.2f} (p90 {np.percentile(ep_b, 90):.2f}) us, counted wait {np.median(ep_c):.2f} (p90 {np.percentile(ep_c, 90):.2f}) us")
