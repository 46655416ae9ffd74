import os, sys
os.environ["DEVIAS_GEMM_DEBUG"] = "10"   # 8 = stamps, 2 = no stores
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch, numpy as np
from devias_amd import ops as o, _lib
M, N, K = 50176, 2304, int(sys.argv[1]) if len(sys.argv) > 1 else 64
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(1764 * 6, dtype=torch.int64, device="cuda")
def call():
    g = _lib.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K; g.lda, g.ldb, g.ldc = K, K, N
    g.dtype = 1; g.split_k = 1; g.ws = ws.data_ptr()
    _lib.check(_lib.load().devias_gemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "gemm")
for _ in range(3): call()
torch.cuda.synchronize()
d = ws.cpu().numpy().reshape(1764, 6)
t0 = d[:, 0].min()
start = (d[:, 0] - t0) / 100.0; issue = (d[:, 1] - d[:, 0]) / 100.0; wait = (d[:, 2] - d[:, 1]) / 100.0; loop = (d[:, 3] - d[:, 2]) / 100.0
end = (d[:, 3] - t0) / 100.0
print(f"K={K}: kernel span {end.max():.1f} us; block start times: p10 {np.percentile(start,10):.1f} p50 {np.percentile(start,50):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f}")
print(f"  per block (us): entry->loads issued  med {np.median(issue):.2f} max {issue.max():.2f}; first wait+barrier med {np.median(wait):.2f} p90 {np.percentile(wait,90):.2f} max {wait.max():.2f}; loop med {np.median(loop):.2f} max {loop.max():.2f}")
order = np.argsort(d[:, 0])
print("  first 12 block starts:", np.round(start[order][:12], 2), " xcc:", d[order][:12, 4] & 0xf)
print("  blocks started in first 2us:", int((start < 2).sum()), " in first 20us:", int((start < 20).sum()))
