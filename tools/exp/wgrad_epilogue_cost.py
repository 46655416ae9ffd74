import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DEVIAS_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdevias_amd_dbg.so")
import torch
from devias_amd import ops as o
M = 50176
for name, No, Ki in (("wfc1", 3072, 768), ("wproj", 768, 768)):
    g = (torch.randn(M, No, device="cuda") * 0.1).bfloat16(); x = (torch.randn(M, Ki, device="cuda") * 0.5).bfloat16()
    out = torch.empty(No, Ki, device="cuda")
    for dbg in (0, 2, 0, 2):
        o.set_option("gemm_debug", dbg)
        for _ in range(3): o.wgrad(g, x, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): o.wgrad(g, x, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"{name} gemm_debug={dbg} ({'no epilogue stores' if dbg else 'full'}): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (+ reduce)")
