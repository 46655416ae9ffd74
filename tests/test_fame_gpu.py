"""FAME on the device (devias_amd.fame.FAME, devias_fame_* in the C ABI) against the reference's outputs (goldens) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from devias_amd import synth
from oracle.fame_cpu import FameOracle, gaussian_blur2d

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_blur_matches_oracle():
    from devias_amd import _lib
    x = torch.rand(5, 70, 97)
    ref = gaussian_blur2d(x[:, None], 11, 11 / 3)[:, 0]
    xd, out = x.cuda(), torch.empty(5, 70, 97, device="cuda")
    _lib.check(_lib.load().devias_fame_blur(xd.data_ptr(), out.data_ptr(), 5, 70, 97, 11, 11 / 3, torch.cuda.current_stream().cuda_stream), "blur")
    assert float((out.cpu() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("name", ["fame_t8", "fame_t16_all"])
def test_fame_matches_reference_golden(name):
    """Masks are top-k selections of blurred fp32 images: a pixel whose value is within rounding of the k-th value may flip, so the
    binary masks must agree on >= 99.9 % of the pixels and the pooled masks to a few 1/256 steps (mean |diff| < 2e-4); the mixing itself is exact given the mask."""
    from devias_amd.fame import FAME
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    B, T, size = int(fx["B"]), int(fx["T"]), int(fx["size"])
    x = synth.scene_video(B, T, size)
    f = FAME(beta=float(fx["beta"]), prob_aug=float(fx["prob_aug"]))
    assert "FAME" in str(f)
    label = torch.from_numpy(fx["label"])
    vids, lab, (m, mpf) = f(x.cuda(), label.cuda(), index=torch.from_numpy(fx["perm"]), rand_batch=torch.from_numpy(fx["rand"]))
    assert np.array_equal(lab.cpu().numpy(), fx["out_label"])
    assert m.shape == fx["mask"].shape and mpf.shape == fx["masks_per_frame"].shape
    for got, want in ((m.cpu().numpy(), fx["mask"]), (mpf.cpu().numpy(), fx["masks_per_frame"])):
        d = np.abs(got - want)
        assert float(d.max()) <= 6 / 256 + 1e-7 and float(d.mean()) < 2e-4, (float(d.max()), float(d.mean()))
    assert abs(float(m.mean()) - float(fx["mask"].mean())) < 1e-6                       # exactly num_fg pixels are selected
    binmask, _, _ = f.masks(x.cuda())
    ref_bits = np.unpackbits(fx["binmask"])[: B * size * size].reshape(B, size, size)
    agree = float((binmask[:, 0].cpu().numpy() == ref_bits).mean())
    assert agree >= 0.999, agree
    # mixing: bit-exact against the oracle's formula evaluated with the DEVICE mask
    o = FameOracle(beta=float(fx["beta"]), prob_aug=float(fx["prob_aug"]))
    perm, rand = torch.from_numpy(fx["perm"]), torch.from_numpy(fx["rand"])
    m5 = binmask[:, 0].cpu().float().view(B, 1, 1, size, size)
    fuse = x[perm] * (1 - m5) + x * m5
    if o.prob_aug < 1:
        aug, ori = torch.where(rand < o.prob_aug)[0], torch.where(rand >= o.prob_aug)[0]
        want = torch.cat([fuse[aug], x[ori]], 0)
    else:
        want = fuse
    assert torch.equal(vids.cpu(), want)
    got_sample = vids.cpu().flatten()[torch.from_numpy(fx["video_sample_idx"])].numpy()
    assert float((got_sample == fx["video_sample"]).mean()) >= 0.995


def test_fame_deterministic_and_ties():
    """constant clips: every pixel ties -- the selection must still pick exactly k pixels, in index order, identically twice"""
    from devias_amd.fame import FAME
    x = torch.zeros(2, 3, 4, 64, 64).cuda()
    f = FAME(beta=0.25, prob_aug=1.0)
    b1, p1, _ = f.masks(x)
    b2, p2, _ = f.masks(x)
    assert torch.equal(b1, b2) and torch.equal(p1, p2)
    assert int(b1[0, 0].sum()) == int(0.25 * 64 * 64)
    assert bool(b1[0, 0].flatten()[: int(0.25 * 64 * 64)].all())


def test_engine_step_with_fame():
    """engine.train_one_epoch with mask_model=FAME (engine_for_slot.py:106-108): one step runs and yields finite loss"""
    from devias_amd.fame import FAME
    import devias_amd
    from devias_amd.engine_for_slot import train_one_epoch
    from devias_amd.optim import FusedAdamW
    from devias_amd.train_loss import TrainLoss
    model = devias_amd.create_model("slot_vit_small_patch16_224", num_classes=400, all_frames=4, num_latents=2, slot_matching_method="matching",
                                    agg_weights_tie=True, agg_depth=2, num_scene_classes=365, compute_dtype="bf16")
    synth.fill_module_(model, seed=0)
    model = model.cuda()
    B = 4
    x = synth.scene_video(B, 4, 224)
    y = synth.targets(B, 400)
    tl = synth.teacher_logits(B, 365).cuda()
    crit = TrainLoss(scene_criterion="KL", num_action_classes=400, slot_matching_method="matching", scene_loss_weight=4000,
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0)
    opt = FusedAdamW(model.parameters(), lr=1e-4)
    st = train_one_epoch(model, tl, crit, [(x, y)], opt, "cuda", 0, max_norm=1.0, mask_model=FAME(beta=0.5, prob_aug=0.5), check_finite_every=1)
    assert np.isfinite(st["loss"]) and st["grad_norm"] > 0


@pytest.mark.parametrize("B,T,size,beta", [(2, 4, 96, 0.5), (3, 6, 160, 0.25), (1, 2, 224, 0.7)])
def test_fame_masks_match_oracle_other_geometries(B, T, size, beta):
    """sizes other than the goldens' (image not a multiple of the 32-pixel blur tile, 2-frame clips, other foreground fractions): device
    masks vs the CPU oracle on the same structured clips"""
    from devias_amd.fame import FAME
    x = synth.scene_video(B, T, size, seed=2100 + size)
    o = FameOracle(beta=beta, prob_aug=1.0)
    _, _, (om, ompf), (obin, _, _) = o.forward(x, torch.arange(B), torch.arange(B).flip(0), torch.zeros(B))
    f = FAME(beta=beta, prob_aug=1.0)
    binmask, pooled, pooled_pf = f.masks(x.cuda())
    assert int(binmask[:, 0].sum()) == B * int(beta * size * size)
    agree = float((binmask[:, 0].cpu().float() == obin).float().mean())
    assert agree >= 0.995, agree
    assert float((pooled.cpu() - om).abs().mean()) < 2e-3
    assert float((pooled_pf.reshape(B, -1).cpu() - ompf).abs().mean()) < 2e-3


def test_device_kernels_match_hand_computed_kornia_vectors():
    """the HIP blur and colour-bin kernels against the float64 known-answer vectors of tests/golden/make_kornia_vectors.py (kornia's
    published GaussianBlur2d / rgb_to_hsv formulas), not only against the oracle's restatement of them"""
    import json
    from devias_amd import _lib
    kv = json.load(open(os.path.join(GOLD, "kornia_vectors.json")))
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    imgs, want = [], []
    for case in kv["blur_cases"]:
        H, W = case["shape"]
        if "impulse" in case:
            img = torch.zeros(H, W)
            img[case["impulse"][0], case["impulse"][1]] = 1.0
        else:
            img = torch.tensor(case["image"], dtype=torch.float32)
        imgs.append(img); want.append(torch.tensor(case["blurred"], dtype=torch.float64))
    x = torch.stack(imgs).cuda().contiguous()
    out = torch.empty_like(x)
    _lib.check(lib.devias_fame_blur(x.data_ptr(), out.data_ptr(), x.shape[0], 16, 16, kv["ksize"], kv["sigma"], st), "blur")
    assert float((out.cpu().double() - torch.stack(want)).abs().max()) < 5e-7
    # colour bins: a clip that is constant in time, one listed colour per pixel, ImageNet-normalised as the data loader delivers it
    cases = [c for c in kv["hsv_cases"] if c["bin_margin"] > 5e-3]
    n = len(cases)
    Wd = 16
    rgb = torch.tensor([c["rgb"] for c in cases], dtype=torch.float32)
    pad = rgb[:1].repeat(Wd - n % Wd if n % Wd else 0, 1)
    px = torch.cat([rgb, pad]).t().reshape(3, 1, -1, Wd)                      # [3, 1, H, W]
    mean = torch.tensor(synth.IMAGENET_MEAN).view(3, 1, 1, 1)
    std = torch.tensor(synth.IMAGENET_STD).view(3, 1, 1, 1)
    clip = ((px - mean) / std).repeat(1, 2, 1, 1)[None].contiguous().cuda()   # [1, 3, T=2, H, W]
    H = clip.shape[3]
    diffs = torch.empty(2, H, Wd, device="cuda")
    cmap = torch.empty(1, H * Wd, dtype=torch.int16, device="cuda")
    _lib.check(lib.devias_fame_diff_color(clip.data_ptr(), 1, 2, H, Wd, diffs.data_ptr(), cmap.data_ptr(), st), "diff_color")
    got = cmap.cpu().reshape(-1)[:n].tolist()
    assert got == [c["fame_bin"] for c in cases], (got, [c["fame_bin"] for c in cases])
    assert float(diffs.abs().max()) == 0.0                                     # identical frames: no motion
