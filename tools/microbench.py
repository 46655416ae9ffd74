#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (hipEvents on torch's current stream, which is the stream the kernels use)."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devias_amd import ops as o


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    B, N, D, H = 32, 1568, 768, 12
    M = B * N
    res = {}
    bf = torch.bfloat16
    x = torch.randn(M, D, device=dev).to(bf)
    for name, n_out, k in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D), ("agg_kv", 4096, D), ("patch", D, 1536)):
        a = torch.randn(M, k, device=dev).to(bf)
        w = (torch.randn(n_out, k, device=dev) * 0.02).to(bf)
        dy = torch.randn(M, n_out, device=dev).to(bf)
        fl = 2.0 * M * n_out * k
        t = timeit(lambda: o.gemm(a, w)); res[f"gemm_fwd_{name}"] = (t, fl / t / 1e9)
        t = timeit(lambda: o.gemm(dy, w, trans_b=True)); res[f"gemm_dgrad_{name}"] = (t, fl / t / 1e9)
        t = timeit(lambda: o.wgrad(dy, a)); res[f"gemm_wgrad_{name}"] = (t, fl / t / 1e9)
    qkv = torch.randn(M, 3 * D, device=dev).to(bf)
    out, lse = o.mhsa_fwd(qkv, B, N, H, 0.125)
    do = torch.randn_like(out)
    fl = 4.0 * B * H * N * N * 64
    t = timeit(lambda: o.mhsa_fwd(qkv, B, N, H, 0.125)); res["mhsa_fwd"] = (t, fl / t / 1e9)
    t = timeit(lambda: o.mhsa_bwd(qkv, out, do, lse, B, N, H, 0.125)); res["mhsa_bwd"] = (t, 2.5 * fl / t / 1e9)
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6)
    by = 2.0 * M * D * 2
    t = timeit(lambda: o.layernorm_fwd(x, g, b, 1e-6)); res["ln_fwd"] = (t, by / t / 1e6)
    t = timeit(lambda: o.layernorm_bwd(y, x, g, mean, rstd, dres=x)); res["ln_bwd"] = (t, 2 * by / t / 1e6)
    dy = torch.randn(M, 4 * D, device=dev).to(bf)
    t = timeit(lambda: o.colsum(dy)); res["colsum_3072"] = (t, M * 4 * D * 2 / t / 1e6)
    S, h, dh = 2, 4, 512
    q = torch.randn(B * S, h * dh, device=dev).to(bf)
    kv = torch.randn(M, 2 * h * dh, device=dev).to(bf)
    A, r, so = o.slot_attn_fwd(q, kv, B, S, N, h, dh, dh ** -0.5)
    t = timeit(lambda: o.slot_attn_fwd(q, kv, B, S, N, h, dh, dh ** -0.5)); res["slot_fwd"] = (t, kv.numel() * 2 / t / 1e6)
    t = timeit(lambda: o.slot_attn_bwd(q, kv, A, r, so, so, None, B, S, N, h, dh, dh ** -0.5)); res["slot_bwd"] = (t, kv.numel() * 2 / t / 1e6)
    L = 8
    st = lambda t_: t_.unsqueeze(0).repeat(L, *([1] * t_.dim())).contiguous()
    qs, dos, dss, As, rs = st(q), st(so), st(A), st(A), st(r)
    t = timeit(lambda: o.slot_attn_kv_grad(qs, dos, dss, As, rs, L, B, S, N, h, dh, dh ** -0.5)); res["slot_kv_grad"] = (t, kv.numel() * 2 / t / 1e6)
    # small-M slot MLP GEMMs
    xs = torch.randn(B * S, D, device=dev).to(bf)
    w1 = (torch.randn(4 * D, D, device=dev) * 0.02).to(bf)
    w2 = (torch.randn(D, 4 * D, device=dev) * 0.02).to(bf)
    hs = torch.randn(B * S, 4 * D, device=dev).to(bf)
    t = timeit(lambda: o.gemm(xs, w1)); res["smallM_ff1"] = (t, 0)
    t = timeit(lambda: o.gemm(hs, w2)); res["smallM_ff2"] = (t, 0)
    for k, (t, r_) in res.items():
        unit = "GB/s" if k.startswith(("ln", "colsum", "slot")) else "TFLOP/s"
        print(f"{k:24s} {t:9.4f} ms   {r_:10.1f} {unit}")
    json.dump({k: {"ms": v[0], "rate": v[1]} for k, v in res.items()}, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    main()
