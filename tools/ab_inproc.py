#!/usr/bin/env python3
"""A/B of process-wide kernel options INSIDE one process: blocks of steps timed with HIP events, the option toggled between blocks in the order A B B A A B B A ...
(no process restarts, no first-run / second-run bias: two separate bench.py runs differ by 0.1-0.3 ms whichever option they carry).
usage: ab_inproc.py [bench.py flags] [hog=K] name=a,b [name=a,b ...]   e.g.  ab_inproc.py --model vit_large gemm_w4=0,1 gemm_tail_split=0,1
hog=K holds K compute units (128 KiB LDS each) on a side stream during every backward of every block that follows on the command line (hog=0 ends it),
as bench.py --cu-hog does: the stand-in for a concurrent collective's kernels."""
import os, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    specs = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a.startswith("--") or ("=" not in a)]     # bench.py flags pass through (--model vit_large --batch 32 ...)
    args = bench.parse()
    dev = torch.device("cuda", 0)
    from devias_amd import synth, ops
    from devias_amd.train_loss import TrainLoss
    model = bench.build_model(args, dev)
    B = args.batch
    N = model.patch_embed.num_patches
    x = synth.video(B, args.frames, args.img_size, seed=1000).to(dev)
    y = synth.targets(B, 400, seed=1000).to(dev)
    tl = synth.teacher_logits(B, 365, seed=1000).to(dev)
    fg = tuple(t.to(dev) for t in synth.fg_masks(B, N, (args.img_size // 16) ** 2, seed=1000))
    crit = TrainLoss(scene_criterion="KL", num_action_classes=400, slot_matching_method="matching", scene_loss_weight=4000,
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0, sync_loss_dict=False)

    from devias_amd import _lib as _dl
    hog = [0]
    hog_us = [36000]                                  # set to 90 % of the backward's device time below (a hold longer than the backward makes the step wait for the hog itself)
    hog_stream = torch.cuda.Stream(device=dev)

    def step():
        for p in model.parameters():
            p.grad = None
        out = model(x)
        total, logits, ld = crit(model, out, (None, tl), y, fg_mask=fg)
        if hog[0]:
            hog_stream.wait_stream(torch.cuda.current_stream(dev))
            _dl.check(_dl.load().devias_debug_cu_hog(hog[0], hog_us[0], hog_stream.cuda_stream), "devias_debug_cu_hog")
        total.backward()
        if hog[0]:
            torch.cuda.current_stream(dev).wait_stream(hog_stream)

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bw = []
    for _ in range(3):
        for p in model.parameters():
            p.grad = None
        total, logits, ld = crit(model, model(x), (None, tl), y, fg_mask=fg)
        eb0.record(); total.backward(); eb1.record(); torch.cuda.synchronize()
        bw.append(eb0.elapsed_time(eb1))
    hog_us[0] = int(0.9 * min(bw) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for spec in specs:
        name, vals = spec.split("=")
        if name == "hog":
            hog[0] = int(vals)
            print(f"--- {hog[0]} CUs held for {hog_us[0] / 1e3:.1f} ms of every backward ({min(bw):.1f} ms un-hogged) from here on", flush=True)
            for _ in range(3):
                step()
            continue
        a, b = (int(v) for v in vals.split(","))
        ts = {a: [], b: []}
        for blk, v in enumerate([a, b, b, a] * 4):
            ops.set_option(name, v)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                step()
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 10)
        ops.set_option(name, b)
        print(f"{name}: {a} -> median {st.median(ts[a]):.3f} ms (min {min(ts[a]):.3f}, max {max(ts[a]):.3f});  {b} -> median {st.median(ts[b]):.3f} ms (min {min(ts[b]):.3f}, max {max(ts[b]):.3f});"
              f"  delta {st.median(ts[b]) - st.median(ts[a]):+.3f} ms", flush=True)


main()
