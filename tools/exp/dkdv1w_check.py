#!/usr/bin/env python3
"""The one-wave-per-SIMD dK / dV kernel (csrc/attn_bwd1w.hip, option attn_dkdv = 1) beside the two-waves-per-SIMD kernel (attn_dkdv = 0) and an fp32 statement of the
same backward: errors of dK / dV / the v_bias gradient per shape (incl. ragged N, N < 256, one head), bitwise run-to-run, and the time of the whole backward
(dQ + dK/dV launches) per option at the step's shape, alternating A B B A.
usage: dkdv1w_check.py [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from devias_amd import ops


def ref_bwd(qkv, d_o, scale):
    """fp32 autograd of softmax(scale q k^T) v on the bf16-rounded inputs"""
    B, N, _, H, Dh = qkv.shape
    x = qkv.float().detach().requires_grad_(True)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    o = torch.softmax((q * scale) @ k.transpose(-1, -2), dim=-1) @ v
    o.permute(0, 2, 1, 3).reshape(B, N, H * Dh).backward(d_o.float())
    return x.grad


def run(qkv, o, d_o, lse, B, N, H, opt, bias):
    ops.set_option("attn_dkdv", opt)        # opt: 0 = the dK / dV kernel of rounds 2-4, 1 = the one-wave-per-SIMD kernel, one workgroup per 256-key block, 2 = the same kernel, persistent workgroups
    if bias:
        dbq = torch.zeros(H * 64, device="cuda"); dbv = torch.zeros(H * 64, device="cuda")
        g = ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))
        return g.view(B, N, 3, H, 64), dbq, dbv
    return ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125).view(B, N, 3, H, 64), None, None


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)).item()


def check():
    torch.manual_seed(0)
    bad = 0
    for (B, N, H) in [(2, 1568, 12), (32, 1568, 12), (1, 256, 1), (1, 64, 2), (2, 33, 3), (1, 300, 8), (2, 100, 3), (1, 1569, 1), (3, 1000, 4), (1, 6400, 2), (2, 257, 12), (1, 1599, 6), (8, 784, 6), (16, 1311, 16),
                      (40, 512, 8), (11, 2048, 24), (40, 288, 16), (3, 1568, 16)]:
        qkv = (torch.randn(B, N, 3, H, 64, device="cuda") * 1.5).to(torch.bfloat16)
        d_o = torch.randn(B, N, H * 64, device="cuda").to(torch.bfloat16)
        o, lse = ops.mhsa_fwd(qkv.view(B * N, 3 * H * 64), B, N, H, 0.125)
        o = o.view(B, N, H * 64)
        ref = ref_bwd(qkv, d_o, 0.125) if B * H * N * N <= 4e8 else None
        for bias in (False, True):
            g0, bq0, bv0 = run(qkv.view(B * N, -1), o.view(B * N, -1), d_o.view(B * N, -1), lse, B, N, H, 0, bias)
            g1, bq1, bv1 = run(qkv.view(B * N, -1), o.view(B * N, -1), d_o.view(B * N, -1), lse, B, N, H, 1, bias)
            g2, bq2, bv2 = run(qkv.view(B * N, -1), o.view(B * N, -1), d_o.view(B * N, -1), lse, B, N, H, 1, bias)
            g3, bq3, bv3 = run(qkv.view(B * N, -1), o.view(B * N, -1), d_o.view(B * N, -1), lse, B, N, H, 2, bias)
            torch.cuda.synchronize()
            same = torch.equal(g1, g2) and torch.equal(g1, g3) and (not bias or (torch.equal(bv1, bv2) and torch.equal(bq1, bq2)))      # run to run, and persistent = one block per workgroup
            fin = bool(torch.isfinite(g1.float()).all())
            e_dq = rel(g1[:, :, 0], g0[:, :, 0]); e_dk = rel(g1[:, :, 1], g0[:, :, 1]); e_dv = rel(g1[:, :, 2], g0[:, :, 2])
            msg = f"B={B} N={N} H={H} bias={int(bias)}: new vs old dQ {e_dq:.1e} dK {e_dk:.1e} dV {e_dv:.1e}"
            if bias:
                msg += f" dbv {rel(bv1, bv0):.1e} dbq {rel(bq1, bq0):.1e}"
                msg += f" | dbv vs sum(dV fp32 ref) {rel(bv1, ref[:, :, 2].sum((0, 1)).reshape(-1)):.1e}" if ref is not None else ""
            if ref is not None:
                msg += f" | vs fp32: old dK {rel(g0[:, :, 1], ref[:, :, 1]):.1e} dV {rel(g0[:, :, 2], ref[:, :, 2]):.1e}; new dK {rel(g1[:, :, 1], ref[:, :, 1]):.1e} dV {rel(g1[:, :, 2], ref[:, :, 2]):.1e}"
            ok = same and fin and e_dq < 2.5e-2 and e_dk < 2.5e-2 and e_dv < 2.5e-2 and (not bias or rel(bq1, bq0) < 5e-3)
            if ref is not None:
                msg += f" dQ old {rel(g0[:, :, 0], ref[:, :, 0]):.1e} new {rel(g1[:, :, 0], ref[:, :, 0]):.1e}"
                ok = ok and rel(g1[:, :, 0], ref[:, :, 0]) < 2.5e-2 and rel(g1[:, :, 1], ref[:, :, 1]) < 2.5e-2 and rel(g1[:, :, 2], ref[:, :, 2]) < 2.5e-2
            bad += 0 if ok else 1
            print(("ok   " if ok else "FAIL ") + msg + ("" if same else " NOT BITWISE run-to-run") + ("" if fin else " NON-FINITE"), flush=True)
    print("ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)
    return bad


def timing():
    B, N, H = 32, 1568, 12
    torch.manual_seed(1)
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
    dbq = torch.zeros(H * 64, device="cuda"); dbv = torch.zeros(H * 64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = {0: [], 1: [], 2: []}
    for opt in [0, 1, 2, 2, 1, 0] * 3:
        ops.set_option("attn_dkdv", opt)
        for _ in range(3):
            ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(12):
            ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))
        e1.record(); torch.cuda.synchronize()
        ts[opt].append(e0.elapsed_time(e1) / 12 * 1e3)
    for opt in (0, 1, 2):
        v = sorted(ts[opt])
        print(f"attn_dkdv={opt}: backward (dQ + dK/dV + bias finish) median {v[len(v) // 2]:.1f} us  (min {v[0]:.1f}, max {v[-1]:.1f}) at B={B} N={N} H={H}", flush=True)


def time_only():
    """one library (DEVIAS_LIB_PATH), attn_dkdv = 1 and 0: for ablation builds, whose results are wrong on purpose"""
    B, N, H = 32, 1568, 12
    torch.manual_seed(1)
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
    dbq = torch.zeros(H * 64, device="cuda"); dbv = torch.zeros(H * 64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for opt in (1, 0):
        ops.set_option("attn_dkdv", opt)
        v = []
        for rep in range(4):
            for _ in range(3):
                ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(12):
                ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125, bias_out=(dbq, dbv))
            e1.record(); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) / 12 * 1e3)
        out.append(sorted(v)[1])
    print(f"{os.environ.get('DEVIAS_LIB_PATH', 'default library')}: backward with attn_dkdv=1 {out[0]:.1f} us, with attn_dkdv=0 {out[1]:.1f} us -> one-wave dK/dV kernel = old kernel {out[0] - out[1]:+.1f} us", flush=True)


def time_three():
    """under rocprofv3: the three dK / dV forms, 40 backward passes each, interleaved (the kernel names tell them apart in the stats)"""
    B, N, H = 32, 1568, 12
    torch.manual_seed(1)
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = ops.mhsa_fwd(qkv, B, N, H, 0.125)
    for rep in range(8):
        for opt in (1, 2, 0):
            ops.set_option("attn_dkdv", opt)
            for _ in range(5):
                ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, 0.125)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if "timeonly3" in sys.argv[1:]:
        time_three(); sys.exit(0)
    if "timeonly" in sys.argv[1:]:
        time_only(); sys.exit(0)
    rc = check()
    if "time" in sys.argv[1:]:
        timing()
    sys.exit(1 if rc else 0)
