#!/usr/bin/env python3
"""RETIRED with its kernel (tools/exp/attn_fwd1w.hip.txt: correct, 379 us against 287 per layer -- DESIGN.md section 5 round 5).  To run it again: copy the kernel back to
devias_amd/csrc/attn_fwd1w.hip, apply tools/exp/attn_fwd1w_dispatch.patch (build list + dispatch + option attn_fwd), add ("attn_fwd1w", "mhsa_fwd1w_kernel", 16, 96) to the
ISA audit's parameters in tests/test_build_cpu.py (count 8 + 8 * 16 + 8 MFMAs, one instantiation), rebuild.
The one-wave-per-SIMD attention forward (csrc/attn_fwd1w.hip, option attn_fwd = 1) beside the four-waves-per-SIMD kernel (attn_fwd = 0) and an fp32 statement of the
same forward: errors of O and lse per shape (incl. N with a rest of queries, N < 256 and N % 32 != 0, which fall back), bitwise run to run, and the time per call at the
step's shape, alternating A B B A.   usage: fwd1w_check.py [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops


def ref_fwd(qkv, B, N, H, scale):
    x = qkv.float().view(B, N, 3, H, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    s = (q * scale) @ k.transpose(-1, -2)
    return (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64), torch.logsumexp(s, dim=-1)


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)).item()


def check():
    torch.manual_seed(0)
    bad = 0
    for (B, N, H, amp) in [(2, 1568, 12, 1.5), (32, 1568, 12, 1.5), (1, 256, 1, 1.5), (1, 64, 2, 1.5), (2, 288, 3, 1.5), (1, 512, 8, 4.0), (2, 100, 3, 1.5), (1, 1569, 1, 1.5), (3, 1024, 4, 6.0),
                           (1, 6400, 2, 1.5), (2, 320, 12, 1.5), (1, 1600, 6, 0.2), (8, 800, 6, 1.5), (5, 1568, 16, 3.0)]:
        qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * amp).to(torch.bfloat16)
        if amp == 6.0:                                   # a maximum that keeps growing along the keys: the rescale path many times per block
            qkv.view(B, N, 3, H, 64)[:, :, 1] *= torch.linspace(0.1, 1.0, N, device="cuda").view(1, N, 1, 1).to(torch.bfloat16)
        ops.set_option("attn_fwd", 0); o0, l0 = ops.mhsa_fwd(qkv, B, N, H, 0.125)
        ops.set_option("attn_fwd", 1); o1, l1 = ops.mhsa_fwd(qkv, B, N, H, 0.125); o2, l2 = ops.mhsa_fwd(qkv, B, N, H, 0.125)
        torch.cuda.synchronize()
        same = torch.equal(o1, o2) and torch.equal(l1, l2)
        fin = bool(torch.isfinite(o1.float()).all()) and bool(torch.isfinite(l1).all())
        msg = f"B={B} N={N} H={H} amp={amp}: new vs old O {rel(o1, o0):.1e} lse {rel(l1, l0):.1e}"
        ok = same and fin and rel(o1, o0) < 2e-2 and rel(l1, l0) < 3e-3
        if B * H * N * N <= 4e8:
            ro, rl = ref_fwd(qkv, B, N, H, 0.125)
            msg += f" | vs fp32: old O {rel(o0, ro):.1e} lse {rel(l0, rl):.1e}; new O {rel(o1, ro):.1e} lse {rel(l1, rl):.1e}"
            ok = ok and rel(o1, ro) < 2e-2 and rel(l1, rl) < 3e-3
        bad += 0 if ok else 1
        print(("ok   " if ok else "FAIL ") + msg + ("" if same else " NOT BITWISE run-to-run") + ("" if fin else " NON-FINITE"), flush=True)
    print("ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)
    return bad


def timing():
    B, N, H = 32, 1568, 12
    torch.manual_seed(1)
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = {0: [], 1: []}
    for opt in [0, 1, 1, 0] * 3:
        ops.set_option("attn_fwd", opt)
        for _ in range(3):
            ops.mhsa_fwd(qkv, B, N, H, 0.125)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.mhsa_fwd(qkv, B, N, H, 0.125)
        e1.record(); torch.cuda.synchronize()
        ts[opt].append(e0.elapsed_time(e1) / 20 * 1e3)
    for opt in (0, 1):
        v = sorted(ts[opt])
        print(f"attn_fwd={opt}: forward median {v[len(v) // 2]:.1f} us  (min {v[0]:.1f}, max {v[-1]:.1f}) at B={B} N={N} H={H}", flush=True)


if __name__ == "__main__":
    rc = check()
    if "time" in sys.argv[1:]:
        timing()
    sys.exit(1 if rc else 0)
