#!/usr/bin/env python3
"""Which torch (ATen) operators run device kernels inside the measured step -- the glue between the library's fused regions: copies, indexing, fills -- with their input shapes.
usage: python tools/torch_ops_in_step.py [bench.py flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity


def main():
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

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    K = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        for _ in range(K):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = getattr(e, "self_cuda_time_total", 0)
        if dt > 0 and e.key.startswith("aten::"):
            rows.append((dt / K, e.count / K, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    print(f"ATen operators with device time inside the step (per step, {K} steps profiled):")
    for dt, cnt, key, shp in rows[:40]:
        print(f"  {dt:8.1f} us  x{cnt:5.1f}  {key:28s} {shp}")
    print(f"  total {sum(r[0] for r in rows):.1f} us per step in {sum(r[1] for r in rows):.0f} operator calls")


main()
