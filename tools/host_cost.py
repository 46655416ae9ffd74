"""Where the host time of a step goes: enqueue time of ONE step on an idle queue (no throttling by a full launch queue), with the fused
regions and with the per-kernel path, plus a cProfile of the region path.  python tools/host_cost.py [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a != "--profile"]
    prof = "--profile" in os.sys.argv
    args = bench.parse()
    dev = torch.device("cuda", 0)
    from devias_amd import synth, _lib
    import devias_amd.modeling_slot as ms
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

    for regions in (True, False):
        ms._REGIONS = regions
        for _ in range(3):
            step()
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            _lib.CALLS[0] = 0
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        ts.sort()
        print(f"regions={regions}: host enqueue of one step on an idle queue: median {ts[len(ts)//2]*1e3:.2f} ms, min {ts[0]*1e3:.2f} ms, library calls {_lib.CALLS[0]}")
    ms._REGIONS = True
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(3):
        step()
        torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(35)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
