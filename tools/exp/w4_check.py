#!/usr/bin/env python3
"""four-wave persistent GEMM (option gemm_w4) against the eight-wave one: bitwise comparison and timing on the block's forward and dgrad shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops as o
from tools.microbench import timeit

dev = "cuda"; bf = torch.bfloat16
M = int(os.environ.get("M", 50176))
sk = int(os.environ.get("SK", 0))
torch.manual_seed(0)
tot = [0.0, 0.0]
for name, N, K, epi, tb in (("qkv bias", 2304, 768, "bias", 0), ("proj bias+res", 768, 768, "res", 0), ("fc1 bias+gelu+aux", 3072, 768, "gelu", 0),
                            ("fc2 bias+res", 768, 3072, "res", 0), ("fc2 bias+res+rowscale", 768, 3072, "res_rs", 0),
                            ("dfc2 dgelu+colsum", 3072, 768, "dgelu", 1), ("dfc1", 768, 3072, "none", 1), ("dproj", 768, 768, "none", 1),
                            ("dqkv", 768, 2304, "none", 1), ("dqkv colsum", 768, 2304, "cs", 1)):
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(K, N, device=dev) * 0.02).to(bf) if tb else (torch.randn(N, K, device=dev) * 0.02).to(bf)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(bf) if epi.startswith("res") else None
    pre = torch.randn(M, N, device=dev).to(bf) if epi == "dgelu" else None
    rs = (torch.rand(M // 1568, device=dev) > 0.3).float() / 0.7 if epi == "res_rs" else None
    outs, ts = [None, None], [[], []]
    # the two variants are timed ALTERNATELY (A B A B ..., 6 rounds of 10 launches; medians): timed one after the other the second is up to 8 % faster whichever it is
    def make(w4):
        out = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        aux = torch.full((M, N), float("nan"), device=dev, dtype=bf) if epi == "gelu" else None
        cs = torch.full((N,), float("nan"), device=dev) if epi in ("dgelu", "cs") else None
        if epi == "gelu": fn = lambda: o.gemm(a, w, bias=bias, act=o.ACT_GELU, aux_out=aux, out=out)
        elif epi == "res": fn = lambda: o.gemm(a, w, bias=bias, res=res, out=out)
        elif epi == "res_rs": fn = lambda: o.gemm(a, w, bias=bias, res=res, out=out, row_scale=rs, rows_per_scale=1568)
        elif epi == "dgelu": fn = lambda: o.gemm(a, w, trans_b=True, act=o.ACT_DGELU, aux_in=pre, out=out, colsum=cs)
        elif epi == "cs": fn = lambda: o.gemm(a, w, trans_b=True, out=out, colsum=cs)
        elif epi == "none": fn = lambda: o.gemm(a, w, trans_b=True, out=out)
        else: fn = lambda: o.gemm(a, w, bias=bias, out=out)
        return fn, (out, aux, cs)
    fns = [make(0), make(1)]
    used = 0
    def select(w4):
        o.set_option("gemm_w4", 15 if w4 else 0)
    for w4 in (0, 1):
        select(w4)
        c0 = o.counters().get("gemm256p", 0)
        fns[w4][0](); torch.cuda.synchronize()
        if w4: used = o.counters().get("gemm256p", 0) - c0
        outs[w4] = tuple(None if t is None else t.clone() for t in fns[w4][1])
    for rnd in range(6):
        for w4 in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            select(w4)
            ts[w4].append(timeit(fns[w4][0], iters=10, warmup=2) * 1e3)
    ts = [sorted(t)[len(t) // 2] for t in ts]
    same = all(x is None or torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
    nan = bool(torch.isnan(outs[1][0].float()).any())
    diff = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
    tot[0] += ts[0]; tot[1] += ts[1]
    print(f"{name:24s} 8 waves {ts[0]:7.1f} us   4 waves {ts[1]:7.1f} us   bitwise equal: {same}  nan: {nan}  max diff {diff:.3e}  (persistent launches in the 4-wave run: {used})", flush=True)
print(f"sum {tot[0]:.1f} -> {tot[1]:.1f} us")
