#!/usr/bin/env python3
"""Per-shape ledger of the encoder block's GEMMs from a rocprofv3 kernel trace of bench.py.

The kernels of an encoder block run in a fixed order (csrc/regions.hip), so the launches of one kernel template cycle through the shapes it serves:
forward   gemm256p<false, 0>: qkv, fc1 (+GELU, saved pre-activation)        gemm256p<false, 1>: patch embed (once per step), then proj, fc2 (+residual)
backward  gemm256p<true, 2>: dfc2 (+dGELU, column sums)                      gemm256p<true, 0>: dfc1, dproj, dqkv
          gemm256<true, true>: wfc2, wfc1, wproj, wqkv (fp32 split-K slabs; the reduce that follows each is listed beside it)

For every shape: median in-step duration, TFLOP/s, tiles, rounds of 256 CUs, the K-loop model (K-tiles x slope x rounds; slope = the kernel's measured
time per 64-deep K-tile, 1.55 us, tools/gemm_pstamps.py) and the residual = what the launch spends outside its K loops (epilogue intervals, the partial
last round, ramp).  With a second argument (the text written by tools/vendor_gemm_ref.py --names) the vendor library's time and kernel name for the same
shape stand beside it: calibration only, nothing in the product links it.

usage: gemm_ledger.py <trace dir> [vendor file] [--json out.json] [--M rows]"""
import csv, glob, json, re, statistics as st, sys

KT_US = 1.55          # one 256 x 256 x 64 K-tile on every CU (the eight-wave K loop, in-kernel stamps)
CUS = 256


def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def seq(rows, key, grid=None):
    return [dur(r) for r in rows if key in r["Kernel_Name"] and (grid is None or int(r["Grid_Size_X"]) == grid)]


def med(v):
    return st.median(v) if v else float("nan")


def cycle(d, n, drop_first_per=None):
    return [med(d[i::n]) for i in range(n)]


def shapes(rows, M, D=768, depth=12):
    """name -> (M, N, K, kind, median us, launches)"""
    out = {}
    def put(name, m, n, k, kind, d):
        out[name] = dict(M=m, N=n, K=k, kind=kind, us=med(d), n=len(d))
    # the instantiation names carry the epilogue code (EPI of epilogue_swap): <B k-strided, side rows, dynamic queues, EPI>
    def fam(tb, side, epi):
        return [dur(r) for r in rows if re.search(r"gemm256p_kernel<%s, %d, (?:true|false), %s>" % (tb, side, epi), r["Kernel_Name"])]
    if any("gemm256p_kernel<false, 2" in r["Kernel_Name"] for r in rows):
        # round 6: dgrad on transposed weight copies -- every GEMM of the block is a B-k-contiguous launch with its own epilogue instantiation
        put("qkv fwd", M, 3 * D, D, "fwd", fam("false", 0, 8)); put("fc1 fwd +GELU +pre", M, 4 * D, D, "fwd", fam("false", 0, 25))
        f1 = fam("false", 1, 8)
        per = 2 * depth + 1
        body = [x for i, x in enumerate(f1) if i % per != 0]
        put("proj fwd +res", M, D, D, "fwd", body[0::2]); put("fc2 fwd +res", M, D, 4 * D, "fwd", body[1::2]); put("patch embed +pos", M, D, 1536, "fwd", f1[0::per])
        put("dfc2 dgrad +dGELU +colsum", M, 4 * D, D, "dgrad", fam("false", 2, 68))
        b0 = fam("false", 0, 0)
        put("dfc1 dgrad", M, D, 4 * D, "dgrad", b0[0::2]); put("dqkv dgrad", M, D, 3 * D, "dgrad", b0[1::2])
        put("dproj dgrad", M, D, D, "dgrad", fam("false", 0, 64))
    else:
        f0 = seq(rows, "gemm256p_kernel<false, 0")
        put("qkv fwd", M, 3 * D, D, "fwd", f0[0::2]); put("fc1 fwd +GELU +pre", M, 4 * D, D, "fwd", f0[1::2])
        f1 = seq(rows, "gemm256p_kernel<false, 1")
        # one patch-embed launch per step precedes the 2 x depth launches of the blocks
        per = 2 * depth + 1
        body = [x for i, x in enumerate(f1) if i % per != 0]
        put("proj fwd +res", M, D, D, "fwd", body[0::2]); put("fc2 fwd +res", M, D, 4 * D, "fwd", body[1::2])
        put("patch embed +pos", M, D, 1536, "fwd", f1[0::per])
        b2 = seq(rows, "gemm256p_kernel<true, 2")
        put("dfc2 dgrad +dGELU +colsum", M, 4 * D, D, "dgrad", b2)
        b0 = seq(rows, "gemm256p_kernel<true, 0")
        put("dfc1 dgrad", M, D, 4 * D, "dgrad", b0[0::3]); put("dproj dgrad", M, D, D, "dgrad", b0[1::3]); put("dqkv dgrad", M, D, 3 * D, "dgrad", b0[2::3])
    # weight gradients: per step 4 x depth one-round launches (wfc2, wfc1, wproj, wqkv per block, last block first), then the patch embedding's; the small ones
    # (agg block, head) run fewer than 200 workgroups
    w = [dur(r) for r in rows if "gemm256_kernel<true, true" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 200 * 512]
    per = 4 * depth + 1
    if w and len(w) % per == 0:
        body = [x for i, x in enumerate(w) if i % per != per - 1]
        for i, (nm, n, k) in enumerate((("wfc2 wgrad", D, 4 * D), ("wfc1 wgrad", 4 * D, D), ("wproj wgrad", D, D), ("wqkv wgrad", 3 * D, D))):
            out[nm] = dict(M=n, N=k, K=M, kind="wgrad", us=med(body[i::4]), n=len(body[i::4]))
        r = seq(rows, "splitk_reduce_plain_kernel")
        out["split-K reduce (mean of the 4)"] = dict(M=0, N=0, K=0, kind="reduce", us=med([x for x in r if x > 8.0]), n=len(r))
    return out


def model(s):
    """tiles, rounds, K-loop model and residual of a forward / dgrad launch"""
    tiles = (s["M"] // 256) * (s["N"] // 256)
    rounds = tiles / CUS
    kt = s["K"] // 64
    # per XCD group: ceil(tiles / 8) tiles over 32 workgroups; the tail tiles as halves when at most half of the workgroups would be busy
    q = -(-tiles // 8)
    full, rem = divmod(q, CUS // 8)
    eff = full + (0 if rem == 0 else (0.55 if 2 * rem <= CUS // 8 else 1.0))
    return tiles, rounds, eff, kt * KT_US * eff


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    M = int(sys.argv[sys.argv.index("--M") + 1]) if "--M" in sys.argv else 50176
    rows = load(args[0])
    sh = shapes(rows, M)
    vendor = {}
    if len(args) > 1:
        for line in open(args[1]):
            p = line.rstrip("\n").split("\t")
            if len(p) >= 3 and p[0] in sh:
                vendor[p[0]] = (float(p[1]), p[2])
    print(f"{'shape':28s} {'[M, N, K]':>20s} {'in-step us':>10s} {'TFLOP/s':>8s} {'% peak':>7s} {'tiles':>6s} {'rounds':>7s} {'sched':>6s} {'K-loop us':>10s} {'residual':>9s} {'vendor us':>10s} {'ratio':>6s}")
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    js = {}
    for name, s in sh.items():
        if s["us"] != s["us"]:
            continue
        if s["kind"] == "reduce":
            print(f"{name:28s} {'':>20s} {s['us']:10.1f}")
            continue
        fl = 2.0 * s["M"] * s["N"] * s["K"]
        tf = fl / s["us"] / 1e6
        if s["kind"] != "wgrad":
            tiles, rounds, eff, kl = model(s)
            extra = f"{tiles:6d} {rounds:7.2f} {eff:6.2f} {kl:10.1f} {s['us'] - kl:9.1f}"
        else:
            extra = f"{'':6s} {'1.00':>7s} {'':6s} {'':10s} {'':9s}"
        v = vendor.get(name)
        vs = f"{v[0]:10.1f} {v[0] / s['us']:6.2f}" if v else ""
        print(f"{name:28s} {str([s['M'], s['N'], s['K']]):>20s} {s['us']:10.1f} {tf:8.0f} {100 * tf / 2516.6:7.1f} {extra} {vs}")
        if v:
            print(f"{'':28s}   vendor kernel: {v[1]}")
        if name != "patch embed +pos":
            tot[s["kind"]] += s["us"]
        js[name] = dict(us=round(s["us"], 1), tflops=round(tf), launches=s["n"], **({"vendor_us": v[0], "vendor_kernel": v[1]} if v else {}))
    print(f"per block: forward {tot['fwd']:.1f} us, dgrad {tot['dgrad']:.1f} us, wgrad {tot['wgrad']:.1f} us; x 12 blocks = {12 * sum(tot.values()) / 1e3:.2f} ms per step "
          f"(forward + dgrad {12 * (tot['fwd'] + tot['dgrad']) / 1e3:.2f} ms)")
    if "--json" in sys.argv:
        json.dump(js, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


main()
