#!/usr/bin/env python3
"""Is the step limited by board power / temperature?  Device time of ONE step (HIP events) after an idle pause of P ms, for several P, next to the
back-to-back loop: if steps that follow a pause are shorter, the chip is throttled in steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    from devias_amd import synth
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

    def step():
        for p in model.parameters():
            p.grad = None
        out = model(x)
        total, logits, ld = crit(model, out, (None, tl), y, fg_mask=fg)
        total.backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for pause in (0, 20, 50, 100, 200, 500, 1000, 0):
        ts = []
        for it in range(12):
            if pause:
                torch.cuda.synchronize(); time.sleep(pause / 1e3)
            e0.record(); step(); e1.record()
            if pause or it == 11:
                torch.cuda.synchronize()
                if pause: ts.append(e0.elapsed_time(e1))
        if not pause:
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20): step()
            e1.record(); torch.cuda.synchronize()
            print(f"back to back: {e0.elapsed_time(e1) / 20:.2f} ms per step", flush=True)
        else:
            ts.sort()
            print(f"one step after {pause:5d} ms idle: median {ts[len(ts) // 2]:.2f} ms, min {ts[0]:.2f}, max {ts[-1]:.2f}", flush=True)


main()
