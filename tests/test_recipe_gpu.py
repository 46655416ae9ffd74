"""The reference's UCF-101 / HMDB recipe for this very script and model (docs/TRAIN.md:80-125: `--nb_classes 101 --fc_drop_rate 0.5
--drop_path 0.2`): nn.Dropout on the head's input (model/modeling_slot.py:291,393) with the SAME mask as the oracle, forward and backward;
the head / loss / slot-selection kernels at head width 101 + 365 = 466 (ragged: not a multiple of 8); one fp32 step at nb_classes = 101
against the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from devias_amd import synth
from oracle import ref_cpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("C", [765, 466])
def test_fc_dropout_matches_oracle_with_the_same_mask(dtype, tol, C):
    from devias_amd.modeling_slot import HeadRegionFn
    R, D, h1, h2, G = 6, 768, 512, 256, 196
    shp = {"hw": (C, D), "hb": (C,), "w0": (h1, D), "b0": (h1,), "w2": (h2, h1), "b2": (h2,), "w4": (G, h2), "b4": (G,)}
    P = {k: synth.param_values("fcdrop." + k, s, seed=3) * (0.05 if k.startswith(("hw", "w")) else 0.1) for k, s in shp.items()}
    slots = synth.param_values("fcdrop.slots", (R, D), seed=4)
    keep = 0.5
    mask = ((keep + torch.rand((R, D), generator=torch.Generator().manual_seed(11))).floor() / keep)
    wz, wm = synth.param_values("fcdrop.wz", (R, C), seed=5), synth.param_values("fcdrop.wm", (R, G), seed=6)
    if dtype == torch.bfloat16:                       # the oracle sees the values the kernels see
        slots = slots.bfloat16().float()
        P = {k: (v.bfloat16().float() if v.dim() == 2 else v) for k, v in P.items()}
    # oracle (model/modeling_slot.py:393 with nn.Dropout's mask made explicit; :199-204, :209-216)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    sg = slots.clone().requires_grad_(True)
    Z = F.linear(sg * mask, Pg["hw"], Pg["hb"])
    m = torch.sigmoid(F.linear(F.relu(F.linear(F.relu(F.linear(sg, Pg["w0"], Pg["b0"])), Pg["w2"], Pg["b2"])), Pg["w4"], Pg["b4"]))
    ((Z * wz).sum() + (m * wm).sum()).backward()
    # HIP region
    Pc = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
    sc = slots.to(dtype).cuda().requires_grad_(True)
    Zc, mc = HeadRegionFn.apply(sc, Pc["hw"], Pc["hb"], Pc["w0"], Pc["b0"], Pc["w2"], Pc["b2"], Pc["w4"], Pc["b4"], dtype, mask.cuda().contiguous())
    ((Zc.float() * wz.cuda()).sum() + (mc.float() * wm.cuda()).sum()).backward()
    assert gu.rel(Zc.detach().float().cpu(), Z.detach()) < tol and gu.rel(mc.detach().float().cpu(), m.detach()) < tol
    assert gu.rel(sc.grad.float().cpu(), sg.grad) < tol * 5
    for k in shp:
        assert gu.rel(Pc[k].grad.cpu(), Pg[k].grad) < tol * 5, k
    # dropped elements carry no head gradient: where mask == 0 the slot gradient is the MaskPredictor's alone
    if dtype == torch.float32:
        Pg2 = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        s2 = slots.clone().requires_grad_(True)
        m2 = torch.sigmoid(F.linear(F.relu(F.linear(F.relu(F.linear(s2, Pg2["w0"], Pg2["b0"])), Pg2["w2"], Pg2["b2"])), Pg2["w4"], Pg2["b4"]))
        (m2 * wm).sum().backward()
        z = mask == 0
        assert gu.rel(sc.grad.float().cpu()[z], s2.grad[z]) < 1e-4


def test_fc_dropout_module_switch():
    """masks are drawn in training mode only; eval is deterministic; fc_drop_rate reaches the module as in the reference (:291)"""
    from devias_amd.modeling_slot import VisionTransformer
    m = VisionTransformer(embed_dim=384, num_heads=6, depth=1, qkv_bias=True, num_classes=101, all_frames=2, num_latents=2, agg_weights_tie=True,
                          agg_depth=2, slot_matching_method="matching", fc_drop_rate=0.5, init_scale=1.0, compute_dtype="fp32")
    assert isinstance(m.fc_dropout, torch.nn.Dropout) and m.fc_dropout.p == 0.5 and m.head.weight.shape == (466, 384)
    synth.fill_module_(m, seed=0)
    m = m.cuda()
    x = synth.video(2, 2, 224).cuda()
    m.eval(); a = m(x)[2][0]; b = m(x)[2][0]
    assert torch.equal(a, b)
    m.train(); torch.manual_seed(0); c = m(x); d = m(x)
    assert not torch.equal(c[2][0], d[2][0])                  # slots_head differs between two draws ...
    assert torch.equal(c[2][1], d[2][1]) and torch.equal(c[2][2], d[2][2])      # ... slots and mask predictions do not (un-dropped, :408)
    with pytest.raises(ValueError):
        VisionTransformer(embed_dim=384, num_heads=6, depth=1, drop_rate=1.0)


@pytest.mark.parametrize("dtype,tol_out,tol_grad", [("fp32", 1e-3, 5e-3), ("bf16", 3e-2, None)])
def test_step_at_101_action_classes_vs_oracle(dtype, tol_out, tol_grad):
    """UCF-101 head width 466: fp32 within north_star's 1e-3 of the CPU oracle (logits, loss, every gradient), bf16 bounded"""
    from functools import partial
    from devias_amd.modeling_slot import VisionTransformer
    from devias_amd.train_loss import TrainLoss
    cfg = ref_cpu.SlotViTConfig(embed_dim=384, num_heads=6, depth=2, all_frames=4, num_classes=101, num_latents=2, agg_depth=2, agg_weights_tie=True)
    B = 3
    m = VisionTransformer(patch_size=16, embed_dim=384, depth=2, num_heads=6, mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                          num_classes=101, all_frames=4, tubelet_size=2, init_scale=1.0, num_latents=2, slot_matching_method="matching",
                          agg_weights_tie=True, agg_depth=2, num_scene_classes=365, compute_dtype=dtype)
    synth.fill_module_(m, seed=0)
    m = m.cuda().train()
    x, y, tl, fg = gu.inputs(cfg, B)
    crit = TrainLoss(scene_criterion="KL", num_action_classes=101, slot_matching_method="matching", scene_loss_weight=4000,
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0)
    out = m(x.cuda())
    total, logits, ld = crit(m, out, (None, tl.cuda()), y.cuda(), fg_mask=(fg[0].cuda(), fg[1].cuda()))
    total.backward()
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    ototal, ologits, old, ograds, oout, oidx = ref_cpu.train_step(P, cfg, x, y, tl, fg)
    assert out[2][0].shape == (B * 2, 466)
    e_logits = gu.rel(out[2][0].detach().float().cpu(), oout[2][0].detach())
    e_total = abs(float(total) - float(ototal)) / abs(float(ototal))
    assert e_logits < tol_out and e_total < tol_out, (e_logits, e_total)
    assert crit.last_match[:, 0].cpu().tolist() == oidx[0].tolist() and crit.last_match[:, 1].cpu().tolist() == oidx[1].tolist()
    if tol_grad is not None:
        gmax = max(float(g.abs().max()) for g in ograds.values())
        worst = max(float((p.grad.cpu().double() - ograds[n].double()).abs().max() / max(float(ograds[n].abs().max()), 1e-6 * gmax)) for n, p in m.named_parameters())
        assert worst < tol_grad, worst


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("drop", [False, True])
def test_mlp_head_with_and_without_fc_dropout(dtype, tol, drop):
    """head_type='mlp' (MLPHead fc1 -> ReLU -> fc2, model/modeling_slot.py:23-34) behind nn.Dropout on the head's input (:393), forward and backward against the
    same arithmetic in fp32 autograd with the SAME mask; the mask predictor sees the un-dropped slots (:392)."""
    from devias_amd.modeling_slot import HeadMlpFn
    R, D, C, h1, h2, G = 6, 768, 765, 512, 256, 196
    shp = {"f1w": (512, D), "f1b": (512,), "f2w": (C, 512), "f2b": (C,), "w0": (h1, D), "b0": (h1,), "w2": (h2, h1), "b2": (h2,), "w4": (G, h2), "b4": (G,)}
    P = {k: synth.param_values("mlphead." + k, s, seed=3) * (0.05 if len(s) == 2 else 0.1) for k, s in shp.items()}
    slots = synth.param_values("mlphead.slots", (R, D), seed=4)
    keep = 0.5
    mask = ((keep + torch.rand((R, D), generator=torch.Generator().manual_seed(12))).floor() / keep) if drop else None
    wz, wm = synth.param_values("mlphead.wz", (R, C), seed=5), synth.param_values("mlphead.wm", (R, G), seed=6)
    if dtype == torch.bfloat16:
        slots = slots.bfloat16().float()
        P = {k: (v.bfloat16().float() if v.dim() == 2 else v) for k, v in P.items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    s_ref = slots.clone().requires_grad_(True)
    xin = s_ref * mask if drop else s_ref
    Z = F.linear(F.relu(F.linear(xin, ref["f1w"], ref["f1b"])), ref["f2w"], ref["f2b"])
    m = F.relu(F.linear(s_ref, ref["w0"], ref["b0"])); m = F.relu(F.linear(m, ref["w2"], ref["b2"])); Mk = torch.sigmoid(F.linear(m, ref["w4"], ref["b4"]))
    ((Z * wz).sum() + (Mk * wm).sum()).backward()
    dev = "cuda"
    hp = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
    s_hip = slots.to(dev).to(dtype).requires_grad_(True)
    Zh, Mh = HeadMlpFn.apply(s_hip, hp["f1w"], hp["f1b"], hp["f2w"], hp["f2b"], hp["w0"], hp["b0"], hp["w2"], hp["b2"], hp["w4"], hp["b4"], dtype,
                             mask.to(dev) if drop else None)
    ((Zh.float() * wz.to(dev)).sum() + (Mh.float() * wm.to(dev)).sum()).backward()
    assert gu.rel(Zh.detach().float().cpu(), Z.detach()) < tol and gu.rel(Mh.detach().float().cpu(), Mk.detach()) < tol
    assert gu.rel(s_hip.grad.float().cpu(), s_ref.grad) < tol
    for k in P:
        assert gu.rel(hp[k].grad.float().cpu(), ref[k].grad) < tol, k
