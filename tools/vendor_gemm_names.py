#!/usr/bin/env python3
"""Calibration only (nothing in the product links or calls the vendor library): which kernel the vendor GEMM library behind torch.matmul (hipBLASLt / Tensile) picks for the
encoder block's shapes at M = 50176, and how long it runs -- the kernel NAME encodes macro-tile, wave layout, LDS-DMA, prefetch depth, stream-K.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/vendor_gemm_names.py run      # 12 launches per shape, a marker kernel between the shapes
    python3 tools/vendor_gemm_names.py parse <dir> > vendor.tsv                                           # shape <tab> median us <tab> kernel name(s)

The shape names are tools/gemm_ledger.py's, which prints the vendor's figures beside this library's in-step ones.  Plain epilogues on the vendor side (bias where torch fuses
it): the vendor path would pay for GELU / dGELU / residual / column sums in separate kernels."""
import csv, glob, statistics as st, sys

M, D, F = 50176, 768, 3072
SHAPES = ["qkv fwd", "fc1 fwd +GELU +pre", "proj fwd +res", "fc2 fwd +res", "patch embed +pos", "dfc2 dgrad +dGELU +colsum", "dfc1 dgrad", "dproj dgrad", "dqkv dgrad",
          "wfc2 wgrad", "wfc1 wgrad", "wproj wgrad", "wqkv wgrad"]


def run():
    import torch
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    x, xf, g3, xp = bf(M, D), bf(M, F), bf(M, 3 * D), bf(M, 1536)
    Wqkv, Wp, W1, W2, Wpe = bf(3 * D, D), bf(D, D), bf(F, D), bf(D, F), bf(D, 1536)
    b1, b3, bd = bf(F), bf(3 * D), bf(D)
    lin = torch.nn.functional.linear
    ops = {
        "qkv fwd": lambda: lin(x, Wqkv, b3), "fc1 fwd +GELU +pre": lambda: lin(x, W1, b1), "proj fwd +res": lambda: lin(x, Wp, bd), "fc2 fwd +res": lambda: lin(xf, W2, bd),
        "patch embed +pos": lambda: lin(xp, Wpe, bd), "dfc2 dgrad +dGELU +colsum": lambda: x @ W2, "dfc1 dgrad": lambda: xf @ W1, "dproj dgrad": lambda: x @ Wp,
        "dqkv dgrad": lambda: g3 @ Wqkv, "wfc2 wgrad": lambda: x.t() @ xf, "wfc1 wgrad": lambda: xf.t() @ x, "wproj wgrad": lambda: x.t() @ x, "wqkv wgrad": lambda: g3.t() @ x,
    }
    mark = torch.empty(12345, device="cuda")      # (empty, not zeros: a zeros() is itself a fill kernel of the marker's size)
    for name in SHAPES:
        for _ in range(3):
            ops[name]()
        torch.cuda.synchronize()
        mark.fill_(1.0)                     # the marker: a fill of 12345 floats
        for _ in range(12):
            ops[name]()
        torch.cuda.synchronize()
        mark.fill_(2.0)                     # ... and behind the 12 timed launches (the next shape's warm-up follows)


def parse(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # sections between marker fills (grid of a 12345-element fill: small; identified by name + the 13 + 1 occurrences in order)
    sect, cur = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if "FillFunctor" in n and int(r["Grid_Size_X"]) <= 16384 * 4:
            if cur is not None:
                sect.append(cur)
            cur = []
        elif cur is not None:
            cur.append(r)
    # markers alternate "before the 12 timed launches" / "behind them": the sections between them alternate timed / next shape's warm-up
    sect = sect[0::2]
    for name, s in zip(SHAPES, sect[:len(SHAPES)]):
        per = max(1, len(s) // 12)                   # kernels per matmul call (1, or 2-3 with a stream-K fix-up / a bias kernel)
        timed = s[:12 * per]
        calls = [sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in timed[i * per:(i + 1) * per]) for i in range(12)]
        names = []
        for r in timed[:per]:
            if r["Kernel_Name"] not in names:
                names.append(r["Kernel_Name"])
        print(f"{name}\t{st.median(calls):.1f}\t{' + '.join(names)}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2])
