#!/usr/bin/env python3
"""Benchmark of the DEVIAS slot-ViT training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a torchrun environment: starts its own N ranks as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = student forward (a1-a13) + fused matching loss (a14) + full backward to every parameter gradient
(+ bucketed RCCL gradient all-reduce when N > 1) on ONE batch of 32 synthetic 16x224^2 clips per GPU, bf16 storage with
fp32 accumulation, inputs resident in HBM before the timed region.  Teacher scene logits are an input tensor
(primary metric of SURVEY.md §8d).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2516.6          # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU (dense; MI355X_MICROARCH.md chip table)
TRAIN_GFLOP_PER_CLIP = {           # algorithmic (tied K/V once), BASELINE.md §3
    ("vit_base", 16, 224): 1109.3, ("vit_small", 8, 224): 143.7, ("vit_large", 16, 224): 3617.4, ("vit_base", 32, 320): 7945.6,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--model", default="vit_base", choices=["vit_base", "vit_small", "vit_large"])
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--drop-path", type=float, default=0.0, help="stochastic depth rate (BASELINE.md quotes the metric at 0; the reference's training default is 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)      # (three timed steps; ~4.5 s each for B = 2 clips on 16 cores)
    ap.add_argument("--no-full-step", action="store_true", help="skip the secondary full-step number (teacher fwd + AdamW)")
    ap.add_argument("--no-probes", action="store_true", help="skip the kernel probes (dominant kernel, fc1 forward, sustained MFMA rate): for runs under rocprofv3, whose "
                    "per-step kernel statistics must contain the training steps only")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"], help="wire format of the gradient all-reduce (N > 1)")
    ap.add_argument("--force-gradsync", action="store_true", help="N = 1 only: run the whole gradient-bucket path (hooks, events, side stream) with the "
                    "collective replaced by a same-size device copy on the side stream")
    ap.add_argument("--rccl-world1", action="store_true", help="N = 1 only: a one-rank RCCL process group, every gradient bucket's all-reduce issued through it "
                    "on the side stream (the backend's own streams and Work handles; RCCL refuses two ranks on one GPU, so this is what a 1-GPU box can run of it)")
    ap.add_argument("--cu-hog", type=int, default=0, help="hold this many CUs (128 KiB LDS each) on a side stream during every backward: stand-in for the CUs "
                    "RCCL's kernels occupy at N > 1")
    ap.add_argument("--reserve-cus", type=int, default=0, help="persistent GEMM grids leave this many CUs free (library option gemm_reserve_cus)")
    ap.add_argument("--cu-hog-us", type=int, default=0, help="how long each --cu-hog workgroup holds its CU; 0 (default) = 90 %% of the backward's device time, measured in two un-hogged steps "
                                                               "(a longer hold than the backward makes the step wait for the hog, not for its work: round 6 found the old fixed 36 ms doing that)")
    return ap.parse_args()


def build_model(args, device):
    from devias_amd import create_model, synth
    name = {"vit_base": "slot_vit_base_patch16_224", "vit_small": "slot_vit_small_patch16_224",
            "vit_large": "slot_vit_large_patch16_224"}[args.model]
    model = create_model(name, img_size=args.img_size, num_classes=400, all_frames=args.frames, tubelet_size=2, drop_path_rate=args.drop_path, init_scale=0.001,
                         num_latents=2, head_type="linear", slot_matching="matching", agg_weights_tie=True, agg_depth=8,
                         num_scene_classes=365, compute_dtype=args.dtype)
    synth.fill_module_(model, seed=0)        # formula weights: identical on every rank, no broadcast needed
    return model.to(device).train()


def cpu_baseline(args):
    """The oracle (a PyTorch-CPU restatement pinned to the reference by goldens) timed on this box's host cores on a bounded
    sample of the same workload: B=2 clips, fp32, 1 warm-up + cpu_steps timed steps."""
    from devias_amd import synth
    from oracle import ref_cpu
    kw = {"vit_base": {}, "vit_small": dict(embed_dim=384, num_heads=6), "vit_large": dict(embed_dim=1024, num_heads=16, depth=24)}[args.model]
    cfg = ref_cpu.SlotViTConfig(all_frames=args.frames, img_size=args.img_size, **kw)
    # PyTorch CPU kernels stop scaling (and thrash) far below the 100+ hardware threads of the GPU hosts; 16 threads is what
    # the reference's CPU path is timed on here -- the count actually used is what `cores` reports
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    B = 2
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    x = synth.video(B, cfg.all_frames, cfg.img_size)
    y, tl, fg = synth.targets(B), synth.teacher_logits(B), synth.fg_masks(B, cfg.num_patches, (args.img_size // 16) ** 2)
    ref_cpu.train_step(P, cfg, x, y, tl, fg)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        ref_cpu.train_step(P, cfg, x, y, tl, fg)
    dt = (time.perf_counter() - t0) / args.cpu_steps
    return {"value": B / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/ref_cpu.py student fwd+loss+bwd, B={B} clips x {args.cpu_steps} timed steps (+1 warm-up), fp32, "
                      f"{args.model} {args.frames}x{args.img_size}^2, {torch.get_num_threads()} of {os.cpu_count()} host cores (PyTorch CPU kernels stop scaling beyond)"}


PMC_PROFILE = "profiles/r6_pmc/summary.json"      # written by tools/runs/r6fin.sh (tools/pmc_summary.py --json), keyed by the kernel-source hash


def pmc_traffic(kernel_substr: str) -> dict:
    """HBM-side bytes per launch of a kernel from the committed PMC passes (PMC_PROFILE: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes over tools/pmc_probe.py, 2 x FETCH + WRITE with the gfx950 correction) -- quoted only while the profile was taken on the kernel sources this run
    uses (devias_amd.build.source_hash); otherwise null with the reason (no literal that a kernel change leaves stale)."""
    path = os.path.join(ROOT, *PMC_PROFILE.split("/"))
    try:
        from devias_amd import build
        prof = json.load(open(path))
        cur = build.source_hash()
        hit = [(k, v) for k, v in prof["kernels"].items() if kernel_substr in k and "traffic_bytes" in v]
        if not hit:
            return {"traffic": None, "traffic_note": f"{kernel_substr} is not in {PMC_PROFILE}"}
        if prof.get("source_hash") != cur:
            return {"traffic": None, "traffic_note": f"{PMC_PROFILE} was taken on kernel sources {prof.get('source_hash')}, this run uses {cur}: "
                                                     f"stale ({hit[0][1]['traffic_bytes']:.4g} B then); re-run tools/runs/r6fin.sh"}
        k, v = hit[0]
        return {"traffic": v["traffic_bytes"], "traffic_dur_us_under_pmc": v["dur_us"],
                "traffic_source": f"{PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; kernel sources {cur})"}
    except (OSError, ValueError, KeyError) as e:
        return {"traffic": None, "traffic_note": f"no usable PMC profile: {e}"}


def in_step_gemm_ms(step, M: int, N: int, K: int, trans_a: int, trans_b: int):
    """Average duration of the devias_gemm launches of one shape INSIDE a real step (library events around each such launch on the launch stream:
    devias_debug_gemm_timer_*): the kernel with the operands, cache contents and clock the step gives it, not a back-to-back loop on fresh random data."""
    import ctypes
    from devias_amd import _lib as _dl
    lib = _dl.load()
    torch.cuda.synchronize()
    _dl.check(lib.devias_debug_gemm_timer_arm(M, N, K, trans_a, trans_b), "devias_debug_gemm_timer_arm")
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n, ms = ctypes.c_int32(0), ctypes.c_float(0.0)
        _dl.check(lib.devias_debug_gemm_timer_read(ctypes.byref(n), ctypes.byref(ms)), "devias_debug_gemm_timer_read")
    finally:
        lib.devias_debug_gemm_timer_arm(0, 0, 0, 0, 0)
    return (ms.value / n.value, n.value) if n.value else (None, 0)


def in_step_gemm_each(step, M: int, N: int, K: int, trans_a: int, trans_b: int):
    """The devias_gemm launches of one signature inside ONE real step, one by one in launch order (milliseconds): forward launches come first.  Needed since round 6: a
    dgrad GEMM on a transposed weight copy has the signature of a forward GEMM (fc1 / dfc2, proj / dproj, fc2 / dfc1)."""
    import ctypes
    from devias_amd import _lib as _dl
    lib = _dl.load()
    torch.cuda.synchronize()
    _dl.check(lib.devias_debug_gemm_timer_arm(M, N, K, trans_a, trans_b), "devias_debug_gemm_timer_arm")
    try:
        step()
        torch.cuda.synchronize()
        n, each = ctypes.c_int32(0), (ctypes.c_float * 64)()
        _dl.check(lib.devias_debug_gemm_timer_read_each(ctypes.byref(n), each, 64), "devias_debug_gemm_timer_read_each")
    finally:
        lib.devias_debug_gemm_timer_arm(0, 0, 0, 0, 0)
    return [each[i] for i in range(n.value)]


def gemm_shapes_probe(args, step) -> dict:
    """`roofline.gemm_shapes`: the encoder block's GEMMs, each timed INSIDE a real step (one armed step per signature): median us over the blocks, TFLOP/s
    and the fraction of the nominal bf16 peak.  The same shapes beside the vendor library's times and kernel names: profiles/r6_gemm_shapes.txt (tools/gemm_ledger.py)."""
    import statistics as st
    D = {"vit_base": 768, "vit_small": 384, "vit_large": 1024}[args.model]
    depth = {"vit_base": 12, "vit_small": 12, "vit_large": 24}[args.model]
    M = args.batch * (args.frames // 2) * (args.img_size // 16) ** 2
    out = {}

    def put(name, n_out, k_in, us):
        fl = 2.0 * M * n_out * k_in
        out[name] = {"us": round(us, 1), "tflops": round(fl / us / 1e6, 1), "frac": round(fl / us / 1e6 / PEAK_BF16_TFLOPS, 4)}

    def med(v):
        return st.median(v) if v else float("nan")
    # (signature, names in launch order within a step: `depth` forward launches, then `depth` backward launches on the transposed weight copy)
    for (n, k), (fwd, bwd, bn, bk) in {(3 * D, D): ("qkv fwd", None, 0, 0), (4 * D, D): ("fc1 fwd +GELU +pre", "dfc2 dgrad +dGELU +colsum", 4 * D, D),
                                       (D, D): ("proj fwd +res", "dproj dgrad +colsum", D, D), (D, 4 * D): ("fc2 fwd +res", "dfc1 dgrad", D, 4 * D),
                                       (D, 3 * D): ("dqkv dgrad", None, 0, 0)}.items():
        ms = in_step_gemm_each(step, M, n, k, 0, 0)
        if len(ms) == depth and bwd is None:
            put(fwd, n, k, med(ms) * 1e3)
        elif len(ms) == 2 * depth and bwd is not None:
            put(fwd, n, k, med(ms[:depth]) * 1e3); put(bwd, bn, bk, med(ms[depth:]) * 1e3)
        else:
            out[fwd] = {"note": f"{len(ms)} launches of signature [{M}, {n}, {k}] in the step: not the {depth} / {2 * depth} this table assumes (dgrad without transposed weight copies?)"}
    for name, n_out, k_in in (("wfc2 wgrad", D, 4 * D), ("wfc1 wgrad", 4 * D, D), ("wproj wgrad", D, D), ("wqkv wgrad", 3 * D, D)):
        ms = in_step_gemm_each(step, n_out, k_in, M, 1, 1)
        if ms:
            put(name + " (+ split-K reduce)", n_out, k_in, med(ms) * 1e3)
    return out


def sustained_mfma_probe(device) -> dict:
    """What the matrix cores of THIS part sustain (VERDICT r4 missing 3): a bare v_mfma_f32_16x16x32_bf16 loop on random bf16 operands held in registers, one
    512-thread workgroup per CU, ~50 ms per launch after a warm launch (devias_debug_mfma_probe); TFLOP/s from HIP events around the launch, the clock the CUs held from
    the kernel's own s_memtime / s_memrealtime stamps (median over workgroups).  The chip lowers its clock under a dense MFMA stream (MI355X_MICROARCH.md, DVFS
    give-back), so this -- not 256 CUs x 2.4 GHz -- is what a perfect kernel could deliver at this power."""
    import ctypes
    from devias_amd import _lib as _dl
    lib = _dl.load()
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    data = torch.randn(1 << 20, device=device).to(torch.bfloat16)
    sink = torch.empty(cus * 512, device=device, dtype=torch.float32)
    stamps = torch.zeros(cus, 2, device=device, dtype=torch.int64)
    st = torch.cuda.current_stream(device).cuda_stream
    run = lambda it: _dl.check(lib.devias_debug_mfma_probe(data.data_ptr(), data.numel(), cus, it, sink.data_ptr(), stamps.data_ptr(), st), "devias_debug_mfma_probe")  # noqa: E731
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(2000); torch.cuda.synchronize()
    e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
    per_iter_ms = e0.elapsed_time(e1) / 2000
    iters = max(2000, int(50.0 / per_iter_ms))
    run(iters)                                           # the clock settles within microseconds; one warm launch of the final length anyway
    e0.record(); run(iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = float(lib.devias_debug_mfma_probe_flops(cus, iters))
    s = stamps.cpu()
    ghz = (s[:, 0].double() / s[:, 1].double().clamp(min=1) * 0.1).median().item()
    return {"tflops": fl / ms / 1e9, "clock_ghz": ghz, "launch_ms": ms, "cus": cus,
            "method": "devias_debug_mfma_probe: bare v_mfma_f32_16x16x32_bf16 loop, random bf16 operands in registers, 8 waves per CU on every CU, "
                      "HIP events around one ~50 ms launch; clock = s_memtime / s_memrealtime x 100 MHz inside the kernel, median over workgroups"}


def trace_top_kernel_probe(args, device, step=None):
    """The kernel at the top of the step's rocprofv3 trace (profiles/*_kernel_stats.csv row 1): the weight-gradient GEMM gemm256_kernel<true, true, 0>
    (both operands k-strided, split-K over M, 16 % of the step), on its largest launch, the fc1 weight gradient dW1 = dY^T X: timed INSIDE real steps
    (`avg_ms`, in_step_gemm_ms) and, for continuity with rounds 1-4, in a back-to-back loop on random operands (`avg_ms_back_to_back`)."""
    from devias_amd import ops
    D = {"vit_base": 768, "vit_small": 384, "vit_large": 1024}[args.model]
    M = args.batch * (args.frames // 2) * (args.img_size // 16) ** 2
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dy = torch.randn(M, 4 * D, device=device).to(dt)
    x = torch.randn(M, D, device=device).to(dt)
    call = lambda: ops.wgrad(dy, x)  # noqa: E731
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    loop_ms = e0.elapsed_time(e1) / n
    ms, n_in_step = in_step_gemm_ms(step, 4 * D, D, M, 1, 1) if step is not None else (None, 0)
    timed = "inside real steps (library events around each launch)" if ms is not None else "back-to-back loop on random operands"
    if ms is None:
        ms = loop_ms
    fl = 2.0 * M * 4 * D * D
    es = 2 if dt == torch.bfloat16 else 4
    probe = {"name": "gemm256_kernel<true, true, 0> 256x256x64 LDS-DMA, split-K over M + fixed-order slab reduce: fc1 weight gradient (row 1 of the step's kernel trace)",
             "flop_per_launch": fl, "avg_ms": ms, "achieved": fl / ms / 1e9, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS,
             "timed": timed, "launches_timed": n_in_step, "avg_ms_back_to_back": loop_ms,
             "avg_ms_includes": "the split-K reduce launch that follows the product",
             "algorithmic_bytes": (M * 4 * D + M * D) * es + 4 * D * D * 4}
    if args.model == "vit_base" and args.batch == 32 and args.frames == 16 and args.dtype == "bf16":
        probe.update(pmc_traffic("gemm256_kernel<true, true, 0"))
    return probe


def fc1_fwd_probe(args, device, step=None):
    """The persistent 256x256 MFMA GEMM on the fc1 launch of this workload's encoder block (the probe rounds 1-4 called `dominant_kernel`): timed INSIDE
    real steps (`avg_ms`) and in the back-to-back loop on random operands of rounds 1-4 (`avg_ms_back_to_back`)."""
    from devias_amd import ops
    D = {"vit_base": 768, "vit_small": 384, "vit_large": 1024}[args.model]
    M = args.batch * (args.frames // 2) * (args.img_size // 16) ** 2
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    a = torch.randn(M, D, device=device).to(dt)
    w = (torch.randn(4 * D, D, device=device) * 0.02).to(dt)
    bias = torch.randn(4 * D, device=device) * 0.02
    pre = torch.empty(M, 4 * D, device=device, dtype=dt)
    call = lambda: ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, aux_out=pre)      # exactly the fc1 launch of the encoder block  # noqa: E731
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    loop_ms = e0.elapsed_time(e1) / n
    ms, n_in_step = None, 0
    if step is not None:
        # (since round 6 the dfc2 dgrad on the transposed weight copy has fc1's signature: the step's first `depth` launches of it are the forward's)
        each = in_step_gemm_each(step, M, 4 * D, D, 0, 0)
        depth = {"vit_base": 12, "vit_small": 12, "vit_large": 24}[args.model]
        fwd = each[:depth] if len(each) >= depth else each
        if fwd:
            ms, n_in_step = sum(fwd) / len(fwd), len(fwd)
    timed = "inside a real step (library events around each launch)" if ms is not None else "back-to-back loop on random operands"
    if ms is None:
        ms = loop_ms
    fl = 2.0 * M * 4 * D * D
    es = 2 if dt == torch.bfloat16 else 4
    probe = {"name": "gemm256p_kernel<false, 0> persistent 256x256x64 LDS-DMA: fc1 forward of the encoder block (bias + GELU + second saved output)",
             "flop_per_launch": fl, "avg_ms": ms, "achieved": fl / ms / 1e9, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS,
             "timed": timed, "launches_timed": n_in_step, "avg_ms_back_to_back": loop_ms,
             "algorithmic_bytes": (M * D + 4 * D * D + 2 * M * 4 * D) * es}
    if args.model == "vit_base" and args.batch == 32 and args.frames == 16 and args.dtype == "bf16":
        probe.update(pmc_traffic("gemm256p_kernel<false, 0, false, 25>"))
    return probe


def rank_proof(device_desc: str) -> dict:
    """What the process group itself says about the job -- the JSON line's `rccl` object at N > 1 (a SCALE record must be able to prove its rank
    count): the collective library's version (RCCL reports itself through torch's nccl bindings), the group's world size and backend, and every rank's device,
    gathered THROUGH the group (a rank that did not join cannot appear).  Collective: every rank calls it."""
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, f"rank {dist.get_rank()}: {device_desc}")
    ver = None
    if dist.get_backend() == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001  (a torch build without the bindings: say so instead of failing the bench)
            ver = f"unavailable: {e}"
    return {"version": ver, "world": dist.get_world_size(), "backend": dist.get_backend(), "devices": names}


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as a CHILD process
    (`python -m torch.distributed.run`, the launch of docs/TRAIN.md:21-24 on one node), BEFORE this process has made any GPU call
    (a process that has initialised the GPU must never exec, and does not need to: the parent only relays).  The child's stdout
    (rank 0's JSON line) and stderr (RCCL banner) are inherited; the parent exits with the child's return code."""
    import socket
    import subprocess
    with socket.socket() as s:                           # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    from devias_amd import synth
    from devias_amd.parallel import GradSync, init_distributed_from_env
    from devias_amd.train_loss import TrainLoss
    rank, local, world = init_distributed_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if world > 1 and os.environ.get("DEVIAS_BENCH_TRACE_LAUNCH"):
        print(f"bench.py rank {rank}/{world} joined the process group ({dist.get_backend()})", file=sys.stderr, flush=True)
        proof = rank_proof(torch.cuda.get_device_name(local % torch.cuda.device_count()) if torch.cuda.is_available() else "no GPU")   # (the plumbing test reads it)
        if rank == 0:
            print("bench.py rccl: " + json.dumps(proof), file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    device = torch.device("cuda", local % torch.cuda.device_count())     # (ranks > GPUs only happens in the gloo plumbing test)
    torch.cuda.set_device(device)

    model = build_model(args, device)
    if args.reserve_cus:
        from devias_amd import ops as _o
        _o.set_option("gemm_reserve_cus", args.reserve_cus)
    if args.cu_hog:                                     # another kernel runs beside backward: what GradSync announces for N > 1
        from devias_amd import ops as _o
        _o.set_option("gemm_concurrent", 1)
    B = args.batch
    N = model.patch_embed.num_patches
    first = rank * B                                   # rank r owns clips [B r, B r + B) of the global batch (weak scaling)
    x = synth.video(B, args.frames, args.img_size, seed=1000, first=first).to(device)
    y = synth.targets(B, 400, seed=1000, first=first).to(device)
    tl = synth.teacher_logits(B, 365, seed=1000, first=first).to(device)
    fg196, fgN = (t.to(device) for t in synth.fg_masks(B, N, (args.img_size // 16) ** 2, seed=1000, first=first))
    crit = TrainLoss(scene_criterion="KL", num_action_classes=400, slot_matching_method="matching", scene_loss_weight=4000,
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0, sync_loss_dict=False)
    comm_dtype = torch.bfloat16 if args.comm_dtype == "bf16" else torch.float32
    if args.rccl_world1 and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    sync = (GradSync(model, comm_dtype=comm_dtype, simulate=args.force_gradsync, collective_at_world1=args.rccl_world1)
            if (world > 1 or args.force_gradsync or args.rccl_world1) else None)
    from devias_amd import _lib as _dl
    hog_stream = torch.cuda.Stream(device=device) if args.cu_hog > 0 else None

    if hog_stream is not None and args.cu_hog_us <= 0:   # size the hold to the backward it stands beside
        eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bw = []
        for _ in range(3):
            for p in model.parameters():
                p.grad = None
            out = model(x)
            total, logits, ld = crit(model, out, (None, tl), y, fg_mask=(fg196, fgN))
            eb0.record(); total.backward(); eb1.record(); torch.cuda.synchronize()
            bw.append(eb0.elapsed_time(eb1))
        args.cu_hog_us = int(0.9 * min(bw) * 1e3)

    def step():
        for p in model.parameters():
            p.grad = None
        out = model(x)
        total, logits, ld = crit(model, out, (None, tl), y, fg_mask=(fg196, fgN))
        if hog_stream is not None:                     # the hog starts when the forward's work is done, i.e. with backward
            hog_stream.wait_stream(torch.cuda.current_stream(device))
            _dl.check(_dl.load().devias_debug_cu_hog(args.cu_hog, args.cu_hog_us, hog_stream.cuda_stream), "devias_debug_cu_hog")
        total.backward()
        if sync is not None:
            sync.finish()
        if hog_stream is not None:
            torch.cuda.current_stream(device).wait_stream(hog_stream)
        return total

    for _ in range(args.warmup):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _dl.CALLS[0] = 0
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        loss = step()
    e1.record()
    host_enqueue = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the device is still running): launch-bound check
    lib_calls = _dl.CALLS[0] / args.steps              # host -> library compute calls per step (one per fused region and direction)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    t = torch.tensor([wall], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t)
    loss_value = float(loss.detach().float().sum())
    # host cost of a step: ONE step enqueued on an idle stream (in the back-to-back loop above the launch queue fills up and the host is throttled
    # to the device's pace, so that number says nothing about the host); median of 5, outside the timed region
    host_idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        host_idle.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    host_idle.sort()
    rccl_info = rank_proof(f"{torch.cuda.get_device_name(device)} (cuda:{device.index})") if world > 1 else None
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    full = None
    if not args.no_full_step and world == 1:
        # secondary number: the same step + frozen teacher forward (a15) + fused AdamW update (a18)
        from devias_amd.modeling_finetune import vit_base_patch16_224
        from devias_amd.optim import FusedAdamW
        if args.model == "vit_base" and args.img_size == 224:
            teacher = vit_base_patch16_224(num_classes=365, all_frames=args.frames, tubelet_size=2, use_mean_pooling=False,
                                           init_scale=0.001, compute_dtype=args.dtype)
            synth.fill_module_(teacher, seed=1)
            teacher = teacher.to(device).eval()
            from devias_amd.fame import FAME
            opt = FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.05)
            fame = FAME(beta=0.5, prob_aug=0.5)
            x32 = x.float()                                   # FAME takes the fp32 clips the data loader delivers

            def full_step():
                for p in model.parameters():
                    p.grad = None
                xs, ys, masks = fame(x32, y)                  # device FAME: masks + fg/bg mixing (engine_for_slot.py:106-108)
                out = model(xs)
                _, tlog = teacher(xs, return_attn=False)
                total, logits, ld = crit(model, out, (None, tlog.float()), ys, fg_mask=masks)
                total.backward()
                opt.step(max_norm=1.0)                        # fused global-norm clip + AdamW, 3 launches
                return total

            for _ in range(2):
                full_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nfs = max(3, args.steps // 2)
            for _ in range(nfs):
                fl = full_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / nfs
            full = {"value": B / dt, "unit": "clips/s", "ms_per_step": dt * 1e3, "steps": nfs,
                    "includes": "FAME masks + clip mixing on device, student fwd+loss+bwd, frozen teacher fwd (1569 tokens), grad-norm clip + fused AdamW (98.4M params)",
                    "final_loss": float(fl.detach().float().sum())}

    clips_per_s = world * B * args.steps / wall
    gflop = TRAIN_GFLOP_PER_CLIP.get((args.model, args.frames, args.img_size))
    ach = clips_per_s / world * gflop / 1e3 if gflop else None       # per-GPU TFLOP/s
    peak_mem = torch.cuda.max_memory_allocated(device) / 2 ** 30
    line = {
        "metric": "clips/sec fwd+bwd, ViT-B/16 16x224^2 slot head, bs=32/GPU" if args.model == "vit_base" and args.frames == 16 and args.img_size == 224 and B == 32
                  else f"clips/sec fwd+bwd, {args.model} {args.frames}x{args.img_size}^2 slot head, bs={B}/GPU",
        "value": clips_per_s, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"slot-{args.model} 16-patch {args.frames}x{args.img_size}^2 ({N} tokens), S=2 slots, tied agg depth 8, "
                               f"B={B} clips/GPU, student fwd + matching loss + bwd" + (f" + RCCL grad all-reduce ({args.comm_dtype} wire format, 64 MiB fp32 buckets, side stream)" if world > 1 else "") +
                               (" + gradient-bucket path with the collective replaced by a device copy (--force-gradsync)" if args.force_gradsync and world == 1 else "") +
                               (f" + gradient-bucket path over a ONE-rank RCCL group ({args.comm_dtype} wire format, --rccl-world1)" if args.rccl_world1 and world == 1 else "") +
                               (f" + stochastic depth {args.drop_path} (--drop-path; NOT the BASELINE configuration)" if args.drop_path else "") +
                               (f" + {args.cu_hog} CUs held on a side stream for {args.cu_hog_us / 1e3:.1f} ms of every backward (--cu-hog)" if args.cu_hog else "") +
                               (f" + persistent GEMM grids sized for {args.reserve_cus} fewer CUs (--reserve-cus)" if args.reserve_cus else ""),
                   "global_batch": world * B, "tokens": N, "parallelism": f"dp{world}", "weights": "formula (devias_amd.synth)",
                   "teacher_logits": "input tensor (primary metric, SURVEY.md 8d)", "optimizer_in_step": False,
                   "kernels": "persistent 256x256 GEMM (forward + dgrad -- the dgrad on transposed bf16 weight copies made once per weight update, both operands k-contiguous; epilogue switches compile-time per call site; "
                              "tail tiles of a partial round split between two workgroups), 32x32x16 MFMA attention forward, two-kernel MFMA attention backward, folded slot "
                              "cross-attention (K/V projections on the slot side); one library call per fused region and direction (csrc/regions.hip)"},
        # schema 5 (round 5): roofline.dominant_kernel = the trace's row 1 (the fc1 weight gradient), probe_fc1_fwd = the fc1 forward, both timed inside real steps;
        # roofline.sustained_peak / frac_of_sustained; `rccl` at N > 1.  schema 4 (round 4): `host_enqueue_ms_per_step` has its round-1/2 meaning again (host time to enqueue the K timed steps back to back: a full launch
        # queue throttles the host to the device's pace, so on a device-bound step it reads ~ the step time); the host's own cost is the idle-stream number
        # schema 6 (round 6): roofline.gemm_shapes = every GEMM of an encoder block timed inside a real step (one armed step per signature)
        "schema": 6,
        "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3, "host_library_calls_per_step": lib_calls,
        "host_idle_enqueue_ms_per_step": host_idle[len(host_idle) // 2] * 1e3,
        "host_idle_enqueue_method": "one step enqueued on an idle stream, median of 5, outside the timed region (python + fused-region library calls + HIP launches)",
        "device_ms_per_step": dev_ms / args.steps, "final_loss": loss_value, "peak_mem_gib": peak_mem,
    }
    if ach is not None and args.no_probes:
        line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                            "scope": f"whole step per GPU: {gflop} algorithmic GFLOP/clip x {B} clips / step time", "note": "--no-probes: kernel probes skipped"}
    elif ach is not None:
        in_step = step if world == 1 else None            # (at N > 1 a step contains collectives: rank 0 cannot run one alone)
        dom = trace_top_kernel_probe(args, device, in_step)
        sus = sustained_mfma_probe(device) if args.dtype == "bf16" else None
        line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_BF16_TFLOPS,
                            "traffic": dom.get("traffic"), "traffic_scope": "per launch of `dominant_kernel` (PMC, see its traffic_source / traffic_note); "
                                                                            "`achieved` / `frac` are the whole step's",
                            "scope": f"whole step per GPU: {gflop} algorithmic GFLOP/clip x {B} clips / step time",
                            # the nominal fraction above stays the headline; beside it the same rate against what this part's matrix cores sustain on random data
                            "sustained_peak": sus, "frac_of_sustained": (ach / sus["tflops"]) if sus else None,
                            "dominant_kernel": dom,                                  # row 1 of the step's rocprofv3 kernel trace
                            "probe_fc1_fwd": fc1_fwd_probe(args, device, in_step),     # (rounds 1-4 reported this one under the key `dominant_kernel`)
                            "gemm_shapes": gemm_shapes_probe(args, in_step) if in_step is not None else None}
    if world > 1:
        # proof of the rank count for a SCALE record: what the process group itself says, and every rank's device
        line["rccl"] = rccl_info
    if full is not None:
        line["full_step"] = full
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    elif world > 1:
        line["cpu_baseline"] = None
        line["cpu_baseline_note"] = "reported by the N = 1 run only (rank 0's host cores are shared with the other ranks' launch threads at N > 1)"
    print(json.dumps(line))
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
