#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE (build container only).

The reference (/root/reference, read-only) is pure Python on PyTorch; timm is not
installed here, so a small in-memory stub provides the four timm symbols it uses
(SURVEY.md §8c / Appendix A).  Weights and inputs come from devias_amd/synth.py
formulae keyed on state_dict names, so only OUTPUTS are committed: per-slot logits,
slot features, mask predictions, the slot attention map, the five loss terms, the
matched slot indices, per-parameter gradient norms + 16 sampled gradient elements
per parameter, and a few intermediate slices for bisecting.

While generating, the CPU oracle (oracle/ref_cpu.py) is checked against the
reference on the same data; a mismatch aborts.  The fixtures are data only.

Usage (in the build container):  python tests/golden/make_goldens.py [--only NAME]
This script never runs on the GPU box and nothing under tests/ imports it.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"

from devias_amd import synth  # noqa: E402
from oracle import ref_cpu  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import dropout_masks  # noqa: E402  (the masks a dropout golden is made with, shared with the tests that replay it)


def install_reference():
    sys.path.insert(0, REF)

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def drop_path(x, drop_prob=0., training=False):  # timm 0.4.12 semantics
        if drop_prob == 0. or not training:
            return x
        keep = 1 - drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        return x.div(keep) * (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()

    reg = {}
    _mod("timm"); _mod("timm.models")
    _mod("timm.models.layers", drop_path=drop_path,
         to_2tuple=lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v),
         trunc_normal_=lambda t, mean=0., std=1., a=-2., b=2.: nn.init.trunc_normal_(t, mean, std, a, b))
    _mod("timm.models.registry", register_model=lambda fn: reg.setdefault(fn.__name__, fn))
    _mod("timm.utils", accuracy=None, ModelEma=object, get_state_dict=lambda m: m.state_dict())
    _mod("tensorboardX", SummaryWriter=object)
    # TrainLoss force-casts the fg masks to fp16 (train_loss.py:136-137) which breaks fp32 backward;
    # masks are k/256 (exact in fp16) so making .half() the identity is value-preserving (SURVEY §8c caveat 1)
    torch.Tensor.half = lambda self: self
    import model.modeling_slot as ms
    import model.modeling_finetune as mf
    from agg_block.agg_block import AggregationBlock
    from utils.loss.train_loss import TrainLoss
    return reg, ms, mf, AggregationBlock, TrainLoss


CONFIGS = {
    # name: (cfg kwargs, batch)
    "vitb_t8": (dict(all_frames=8), 2),
    "vitb_t16": (dict(all_frames=16), 2),
    "vits_t8": (dict(all_frames=8, embed_dim=384, num_heads=6), 2),
    "vitb_t8_s4_untied": (dict(all_frames=8, num_latents=4, agg_weights_tie=False, agg_depth=4), 2),
    "vitb_t8_mlphead": (dict(all_frames=8, head_type="mlp"), 2),          # MLPHead (modeling_slot.py:23-34, 307-313)
    # nn.Dropout inside the encoder (pos_drop :280,356; attn_drop :90,110; proj_drop :92,114; Mlp.drop :58,66) + drop_path, training mode, with the
    # masks of devias_amd.synth.dropout_mask / oracle attn_drop_mask handed to the reference's own modules
    "vitb_t8_dropout": (dict(all_frames=8, _drop=dict(drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1)), 2),
}


def install_masks(model, drops: dict):
    """make the reference's dropout modules multiply by the GIVEN masks instead of drawing their own (instance-level forward overrides)"""
    if "pos" in drops:
        model.pos_drop.forward = lambda x, m=drops["pos"]: x * m.view_as(x)
    for i, blk in enumerate(model.blocks):
        d = drops[i]
        if "attn" in d:
            blk.attn.attn_drop.forward = lambda a, m=d["attn"]: a * m
        if "proj" in d:
            blk.attn.proj_drop.forward = lambda y, m=d["proj"]: y * m.view_as(y)
            blk.mlp.drop.forward = lambda y, m=d["mlp"]: y * m.view_as(y)
        if "path1" in d:
            state = {"n": 0}

            def dp(x, d=d, state=state):           # Block.forward calls self.drop_path twice: attention branch, then MLP branch (:150-151)
                m = d["path1"] if state["n"] % 2 == 0 else d["path2"]
                state["n"] += 1
                return x * m.view(-1, 1, 1)
            blk.drop_path.forward = dp


def build_reference_student(cfg: ref_cpu.SlotViTConfig, reg, ms, AggregationBlock, rates=None):
    if cfg.embed_dim == 768:
        rates = rates or dict(drop_rate=0., attn_drop_rate=0., drop_path_rate=0.)
        return reg["slot_vit_base_patch16_224"](
            num_classes=cfg.num_classes, all_frames=cfg.all_frames, tubelet_size=cfg.tubelet_size,
            drop_path_rate=rates["drop_path_rate"], drop_rate=rates["drop_rate"], attn_drop_rate=rates["attn_drop_rate"], init_scale=1e-3, num_latents=cfg.num_latents, head_type=cfg.head_type,
            slot_matching_method="matching", agg_weights_tie=cfg.agg_weights_tie, agg_depth=cfg.agg_depth,
            num_scene_classes=cfg.num_scene_classes)

    # D != 768: the reference wrapper hard-wires 768 (modeling_slot.py:199,392; agg_block.py:13,16), so compose the
    # reference's OWN parametric classes exactly as VisionTransformer.forward (:379-410) wires them (SURVEY §8c caveat 2)
    D = cfg.embed_dim

    class Composed(nn.Module):
        def __init__(self):
            super().__init__()
            self.patch_embed = ms.PatchEmbed(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=3, embed_dim=D,
                                             num_frames=cfg.all_frames, tubelet_size=cfg.tubelet_size)
            self.pos_embed = ms.get_sinusoid_encoding_table(self.patch_embed.num_patches, D)
            nl = partial(nn.LayerNorm, eps=1e-6)
            self.blocks = nn.ModuleList([ms.Block(dim=D, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
                                                  norm_layer=nl, init_values=0.) for _ in range(cfg.depth)])
            self.norm = nl(D)
            self.agg_block = AggregationBlock(num_latents=cfg.num_latents, weight_tie_layers=cfg.agg_weights_tie,
                                              depth=cfg.agg_depth, input_channels=D, latent_dim=D)
            self.mask_predictor = nn.Module()
            self.mask_predictor.decoder = nn.Sequential(nn.Linear(D, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(),
                                                        nn.Linear(256, 196), nn.Sigmoid())
            self.head = nn.Linear(D, cfg.head_width)

        def forward(self, x):
            x = self.patch_embed(x)
            x = x + self.pos_embed.expand(x.shape[0], -1, -1).type_as(x)
            for blk in self.blocks:
                x = blk(x, return_attn=False)
            x = self.norm(x)
            slots, attn = self.agg_block(x)
            bs, S, _ = slots.size()
            slots = slots.reshape(-1, D)
            slots_head = self.head(slots)
            probs = torch.softmax(slots_head, dim=-1).view(bs, S, -1)
            a_idx = torch.argmax(probs[:, :, :cfg.num_classes].max(dim=-1).values, dim=1)
            s_idx = torch.argmax(probs[:, :, cfg.num_classes:].max(dim=-1).values, dim=1)
            ar = torch.arange(bs)
            sv, hv = slots.view(bs, S, -1), slots_head.view(bs, S, -1)
            mask = self.mask_predictor.decoder(slots)
            mask = mask.squeeze().reshape(slots.shape[0], 196)
            return (sv[ar, a_idx], sv[ar, s_idx]), (hv[ar, a_idx], hv[ar, s_idx], attn), (slots_head, slots, mask)

    return Composed()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def sample_idx(name: str, numel: int, k: int = 16) -> np.ndarray:
    return (synth.hash_u64(7, "gradsample." + name, k) % np.uint64(numel)).astype(np.int64)


def tap_summary(t: torch.Tensor) -> np.ndarray:
    """first 4 tokens x 32 channels of sample 0, then [sum, abs-sum] over everything."""
    head = t[0, :4, :32].reshape(-1).double().numpy()
    return np.concatenate([head, [float(t.double().sum()), float(t.double().abs().sum())]])


def generate(name: str, ref):
    reg, ms, mf, AggregationBlock, TrainLoss = ref
    kw, B = CONFIGS[name]
    kw = dict(kw)
    rates = kw.pop("_drop", None)
    cfg = ref_cpu.SlotViTConfig(**kw)
    torch.manual_seed(0)
    model = build_reference_student(cfg, reg, ms, AggregationBlock, rates)
    model.train()
    drops = None
    if rates is not None:
        assert isinstance(model.pos_drop, nn.Dropout) and model.pos_drop.p == rates["drop_rate"] and model.blocks[0].attn.attn_drop.p == rates["attn_drop_rate"]
        drops = dropout_masks(cfg, B, rates)
        install_masks(model, drops)
    synth.fill_module_(model, seed=0)
    names = [n for n, _ in model.named_parameters()]
    shapes = ref_cpu.param_shapes(cfg)
    assert names == list(shapes.keys()), (set(names) ^ set(shapes.keys()))
    for n, p in model.named_parameters():
        assert tuple(p.shape) == shapes[n], n

    N = cfg.num_patches
    x = synth.video(B, cfg.all_frames, cfg.img_size, seed=1000)
    y = synth.targets(B, cfg.num_classes, seed=1000)
    tl = synth.teacher_logits(B, cfg.num_scene_classes, seed=1000)
    fg = synth.fg_masks(B, N, cfg.grid * cfg.grid, seed=1000)

    taps = {}
    blocks = model.blocks
    blocks[0].register_forward_hook(lambda m, i, o: taps.__setitem__("block0", o.detach()))
    blocks[-1].register_forward_hook(lambda m, i, o: taps.__setitem__(f"block{cfg.depth - 1}", o.detach()))
    model.norm.register_forward_hook(lambda m, i, o: taps.__setitem__("feats", o.detach()))

    crit = TrainLoss(criterion=None, scene_criterion="KL", num_action_classes=cfg.num_classes,
                     slot_matching_method="matching", mask_prediction_loss_weight=1.0,
                     mask_distill_loss_weight=1.0, scene_loss_weight=4000)
    out = model(x)
    total, logits, ld = crit(model, out, (None, tl), y, fg_mask=fg)
    model.zero_grad()
    total.backward()
    (af, sf), (al, sl, attn), (slots_head, slots, maskp) = out
    grads = {n: p.grad.detach() for n, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())

    # ---- oracle vs reference on identical data -------------------------------------------------
    P = synth.fill_params(shapes, seed=0)
    otaps = {}
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oout = ref_cpu.student_forward(Pg, cfg, x, otaps, drops)
    ototal, ologits, old, oidx = ref_cpu.train_loss(cfg, oout, tl, y, fg)
    ototal.backward()
    checks = {
        "slots_head": rel_err(oout[2][0], slots_head), "slots": rel_err(oout[2][1], slots),
        "mask_predictions": rel_err(oout[2][2], maskp), "attn": rel_err(oout[1][2], attn),
        "action_logit": rel_err(oout[1][0], al), "scene_logit": rel_err(oout[1][1], sl),
        "total": rel_err(ototal, total), "logits": rel_err(ologits, logits),
    }
    for k in ld:
        checks["loss." + k] = abs(old[k] - ld[k]) / max(abs(ld[k]), 1e-30)
    # a few gradients are mathematically zero (e.g. the slot-query LayerNorm bias: a shift common to every slot
    # cancels in the slot-axis softmax) and hold only round-off, so errors are scaled by the global gradient maximum too
    gmax = max(float(g.abs().max()) for g in grads.values())
    gerr = {n: float((Pg[n].grad.double() - grads[n].double()).abs().max()
                     / max(float(grads[n].abs().max()), 1e-6 * gmax)) for n in names}
    worst_n = max(gerr, key=gerr.get)
    worst_g = gerr[worst_n]
    print(f"[{name}] worst gradient: {worst_n} err={worst_g:.2e} |g|max={float(grads[worst_n].abs().max()):.3e} global={gmax:.3e}")
    checks["grads(worst)"] = worst_g
    print(f"[{name}] oracle vs reference: " + ", ".join(f"{k}={v:.2e}" for k, v in checks.items()))
    bad = {k: v for k, v in checks.items() if v > (1e-3 if k.startswith("grads") else 5e-5)}
    assert not bad, f"oracle disagrees with the reference: {bad}"

    # matched indices the reference used (recover from the returned logits rows)
    Z = slots_head.view(B, cfg.num_latents, -1)
    i_star = [int(torch.argmin((Z[b] - logits[b]).abs().sum(-1))) for b in range(B)]
    assert i_star == oidx[0].tolist()

    fx = {
        "config": np.array(repr(CONFIGS[name][0])), "batch": np.array(B),
        "slots_head": slots_head.detach().numpy(), "slots": slots.detach().numpy(),
        "mask_predictions": maskp.detach().numpy(), "attn": attn.detach().numpy(),
        "action_feat": af.detach().numpy(), "scene_feat": sf.detach().numpy(),
        "action_logit": al.detach().numpy(), "scene_logit": sl.detach().numpy(),
        "matched_logits": logits.detach().numpy(),
        "match_action_slot": oidx[0].numpy(), "match_scene_slot": oidx[1].numpy(),
        "total_loss": np.array(float(total.detach().double())),
        "loss_names": np.array(list(ld.keys())), "loss_values": np.array([float(ld[k]) for k in ld], dtype=np.float64),
        "param_names": np.array(names),
        "grad_norms": np.array([float(grads[n].double().norm()) for n in names], dtype=np.float64),
        "grad_samples": np.stack([grads[n].reshape(-1)[torch.from_numpy(sample_idx(n, grads[n].numel()))].numpy()
                                  for n in names]),
        "tap_names": np.array(sorted(taps.keys())),
        "taps": np.stack([tap_summary(taps[k]) for k in sorted(taps.keys())]),
    }
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB) total_loss={float(total):.9f} {ld}")


def generate_teacher(ref):
    """Teacher forward golden ("next" row §8f-1): frozen cls-token ViT-B, 8x224^2, B=2."""
    reg, ms, mf, _, _ = ref
    cfg = ref_cpu.SlotViTConfig(all_frames=8)
    torch.manual_seed(0)
    teacher = reg["vit_base_patch16_224"](num_classes=365, all_frames=8, tubelet_size=2, use_mean_pooling=False,
                                          init_scale=1e-3)
    teacher.eval()
    synth.fill_module_(teacher, seed=1)
    shapes = ref_cpu.teacher_param_shapes(cfg)
    names = [n for n, _ in teacher.named_parameters()]
    assert sorted(names) == sorted(shapes.keys()), set(names) ^ set(shapes.keys())
    x = synth.video(2, 8, 224, seed=1000)
    with torch.no_grad():
        tok, logits = teacher(x, return_attn=False)
        P = synth.fill_params(shapes, seed=1)
        otok, ologits = ref_cpu.teacher_forward(P, cfg, x)
    e1, e2 = rel_err(otok, tok), rel_err(ologits, logits)
    print(f"[teacher_vitb_t8] oracle vs reference: token={e1:.2e} logits={e2:.2e}")
    assert e1 < 5e-5 and e2 < 5e-5
    path = os.path.join(ROOT, "tests", "golden", "teacher_vitb_t8.npz")
    np.savez_compressed(path, token=tok.numpy(), logits=logits.numpy())
    print(f"[teacher_vitb_t8] wrote {path}")


def generate_state_dict_keys(ref):
    """state_dict key -> shape of the REAL reference modules (data only): pins checkpoint compatibility (tests/test_checkpoint_cpu.py)."""
    import json
    reg = ref[0]
    out = {}
    m = reg["slot_vit_base_patch16_224"](num_classes=400, all_frames=16, tubelet_size=2, drop_path_rate=0.1, init_scale=1e-3, num_latents=2,
                                         head_type="linear", slot_matching_method="matching", agg_weights_tie=True, agg_depth=8,
                                         num_scene_classes=365)
    out["slot_vit_base_patch16_224.tied_s2_d8"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    m = reg["slot_vit_base_patch16_224"](num_classes=174, all_frames=16, num_latents=4, slot_matching_method="matching",
                                         agg_weights_tie=False, agg_depth=4)
    out["slot_vit_base_patch16_224.untied_s4_d4_nb174"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    t = reg["vit_base_patch16_224"](num_classes=365, all_frames=16, tubelet_size=2, use_mean_pooling=False)
    out["vit_base_patch16_224.cls_365"] = {k: list(v.shape) for k, v in t.state_dict().items()}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reference_state_dict_keys.json"), "w"))
    print("[state_dict_keys]", {k: len(v) for k, v in out.items()})


def generate_optim(ref):
    """Parameter groups (utils/optim_factory.py), cosine schedules and get_grad_norm_ (utils/utils.py) of the REAL reference ->
    tests/golden/optim_factory.json (names, scales and schedule values only)."""
    import contextlib
    import io
    import json
    for mod, names in (("adafactor", ["Adafactor"]), ("adahessian", ["Adahessian"]), ("adamp", ["AdamP"]), ("lookahead", ["Lookahead"]),
                       ("nadam", ["Nadam"]), ("nvnovograd", ["NvNovoGrad"]), ("radam", ["RAdam"]), ("rmsprop_tf", ["RMSpropTF"]),
                       ("sgdp", ["SGDP"])):
        m = types.ModuleType("timm.optim." + mod)
        for n in names:
            setattr(m, n, object)
        sys.modules["timm.optim." + mod] = m
    sys.modules.setdefault("timm.optim", types.ModuleType("timm.optim"))
    import utils.optim_factory as rof
    import utils.utils as ru
    reg = ref[0]
    out = {"groups": {}, "schedules": []}
    models = {
        "tied_s2_d8": reg["slot_vit_base_patch16_224"](num_classes=400, all_frames=8, num_latents=2, slot_matching_method="matching",
                                                       agg_weights_tie=True, agg_depth=8, num_scene_classes=365),
        "untied_s4_d4": reg["slot_vit_base_patch16_224"](num_classes=400, all_frames=8, num_latents=4, slot_matching_method="matching",
                                                         agg_weights_tie=False, agg_depth=4, num_scene_classes=365),
    }
    for mname, m in models.items():
        nl = m.get_num_layers()
        for tag, decay, agg_scale in (("ld0.75", 0.75, 0.1), ("ld0.9_agg0.5", 0.9, 0.5), ("flat", 1.0, 0.1)):
            assigner = rof.LayerDecayValueAssigner([decay ** (nl + 1 - i) for i in range(nl + 2)]) if decay < 1.0 else None
            with contextlib.redirect_stdout(io.StringIO()) as buf:
                rof.get_parameter_groups(m, 0.05, m.no_weight_decay(), assigner.get_layer_id if assigner else None,
                                         assigner.get_scale if assigner else None, agg_block_scale=agg_scale)
            txt = buf.getvalue()
            names = json.loads(txt[txt.index("{"):])
            out["groups"][f"{mname}.{tag}"] = {"num_layers": nl, "layer_decay": decay, "agg_block_scale": agg_scale, "weight_decay": 0.05,
                                               "groups": [[k, v["weight_decay"], v["lr_scale"], v["params"]] for k, v in names.items()]}
    for kw in (dict(base_value=1e-3, final_value=1e-6, epochs=5, niter_per_ep=7, warmup_epochs=2, start_warmup_value=1e-6),
               dict(base_value=0.05, final_value=0.05, epochs=3, niter_per_ep=4),
               dict(base_value=2e-3, final_value=1e-5, epochs=4, niter_per_ep=10, warmup_epochs=1, start_warmup_value=0.0, warmup_steps=13),
               dict(base_value=2e-3, final_value=1e-5, epochs=4, niter_per_ep=10, warmup_epochs=0, warmup_steps=5)):
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                sched = [float.hex(float(v)) for v in ru.cosine_scheduler(**kw)]
            except AssertionError:
                sched = "AssertionError"
        out["schedules"].append({"kwargs": kw, "values": sched})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "optim_factory.json"), "w"))
    print("[optim]", {k: len(v["groups"]) for k, v in out["groups"].items()}, [len(s["values"]) for s in out["schedules"]])


def generate_fame(ref):
    """FAME (utils/transform/fame.py): run the reference's own class -- with oracle/fame_cpu.py's restatements of the two kornia
    functions injected as the `kornia` module (kornia is absent here) and its two random draws replaced by fixed tensors --
    check oracle/fame_cpu.FameOracle against it, and commit the outputs (tests/golden/fame_*.npz)."""
    from oracle import fame_cpu

    class _Blur(nn.Module):
        def __init__(self, ks, sg):
            super().__init__()
            self.ks, self.sg = ks, sg

        def forward(self, x):
            return fame_cpu.gaussian_blur2d(x, self.ks[0], self.sg[0])

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _mod("kornia", filters=_mod("kornia.filters", GaussianBlur2d=_Blur), color=_mod("kornia.color", rgb_to_hsv=fame_cpu.rgb_to_hsv))
    _mod("kornia.augmentation"); _mod("kornia.augmentation.container", VideoSequential=object)
    _mod("torchvision", transforms=_mod("torchvision.transforms")); _mod("torchvision.datasets")
    _mod("torchvision.datasets.video_utils", VideoClips=object)
    import utils.transform.fame as rf
    for name, (B, T, size, beta, prob, rand, perm) in {
        "fame_t8": (3, 8, 224, 0.5, 0.5, [0.2, 0.9, 0.4], [2, 0, 1]),
        "fame_t16_all": (2, 16, 224, 0.3, 1.0, [0.0, 0.0], [1, 0]),
    }.items():
        x = synth.scene_video(B, T, size)
        label = torch.arange(B) * 7 + 1
        perm_t, rand_t = torch.tensor(perm), torch.tensor(rand)
        model = rf.FAME(beta=beta, prob_aug=prob)
        orig = (torch.randperm, torch.rand)
        torch.randperm = lambda n, device=None: perm_t
        torch.rand = lambda n: rand_t
        try:
            with torch.no_grad():
                vids, lab, (m, mpf) = model(x.clone(), label)
        finally:
            torch.randperm, torch.rand = orig
        orc = fame_cpu.FameOracle(beta=beta, prob_aug=prob)
        ov, ol, (om, ompf), (binmask, soft, soft_pf) = orc.forward(x, label, perm_t, rand_t)
        assert torch.equal(ol, lab) and torch.equal(om, m) and torch.equal(ompf, mpf), name
        assert torch.equal(ov, vids), name
        idx = (synth.hash_u64(11, "fame.sample." + name, 4096) % np.uint64(vids.numel())).astype(np.int64)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"),
                            B=B, T=T, size=size, beta=beta, prob_aug=prob, rand=np.array(rand, np.float32), perm=np.array(perm, np.int64),
                            label=label.numpy(), out_label=lab.numpy(), mask=m.numpy(), masks_per_frame=mpf.numpy(),
                            binmask=np.packbits(binmask.numpy().astype(np.uint8)), soft=soft.numpy().astype(np.float16),
                            video_sample_idx=idx, video_sample=vids.flatten()[idx].numpy(),
                            video_sum=np.array([float(vids.double().sum()), float(vids.double().abs().sum())]))
        print(f"[{name}] reference == oracle (bitwise); mask mean {float(m.mean()):.4f}")


def generate_knn(ref):
    """knn_classifier of the reference (utils/eval/run_knn.py:123-163) on formula features -> tests/golden/knn.json."""
    import json
    for m in ("dataset", "dataset.datasets", "dataset.kinetics"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["dataset.datasets"].knn_build_dataset = None
    sys.modules["dataset.kinetics"].VideoClsDataset = object
    import utils.eval.run_knn as rk
    n_train, n_test, D, C = 600, 230, 64, 7
    lab_tr = torch.from_numpy((synth.hash_u64(5, "knn.lab.train", n_train) % np.uint64(C)).astype(np.int64))
    lab_te = torch.from_numpy((synth.hash_u64(5, "knn.lab.test", n_test) % np.uint64(C)).astype(np.int64))
    cent = synth.param_values("knn.centroids.weight", (C, D), seed=5) * 20
    f_tr = torch.nn.functional.normalize(cent[lab_tr] + synth.param_values("knn.noise.train.weight", (n_train, D), seed=5) * 110, dim=1)
    f_te = torch.nn.functional.normalize(cent[lab_te] + synth.param_values("knn.noise.test.weight", (n_test, D), seed=5) * 110, dim=1)
    out = {"n_train": n_train, "n_test": n_test, "D": D, "C": C, "cases": []}
    for k, T in ((20, 0.07), (3, 0.07), (10, 1.0)):
        t1, t5 = rk.knn_classifier(f_tr, lab_tr, f_te, lab_te, k, T, num_classes=C)
        out["cases"].append({"k": k, "T": T, "top1": t1, "top5": t5})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "knn.json"), "w"))
    print("[knn]", out["cases"])


def generate_loss_criteria(ref):
    """The reference's TrainLoss itself (utils/loss/train_loss.py:85-187) on seeded student outputs, for BOTH scene criteria
    ('KL' of the recipes, 'CE' of `run_slot_finetuning.py --scene_criterion CE`): inputs, the five loss terms, the matched logits
    and the gradients with respect to the four differentiable inputs.  The oracle's `train_loss` is checked against it here."""
    _, _, _, _, TrainLoss = ref
    fx = {}
    for S in (2, 3):
        B, nb, ns, D, G, N, nh = 4, 400, 365, 768, 196, 392, 4
        g = torch.Generator().manual_seed(4100 + S)
        base = dict(slots_head=torch.randn(B * S, nb + ns, generator=g) * 2.0, slots=torch.randn(B * S, D, generator=g),
                    maskp=torch.rand(B * S, G, generator=g), attn=torch.rand(B * nh, S, N, generator=g),
                    teacher=torch.randn(B, ns, generator=g) * 3.0, target=torch.randint(0, nb, (B,), generator=g),
                    fg=torch.randint(0, 257, (B, G), generator=g).float() / 256.0, fgN=torch.randint(0, 257, (B, N), generator=g).float() / 256.0)
        for k, v in base.items():
            fx[f"s{S}.{k}"] = v.numpy()
        cfg = ref_cpu.SlotViTConfig(all_frames=8, num_latents=S)
        for crit_name in ("KL", "CE"):
            leaves = {k: base[k].clone().requires_grad_(True) for k in ("slots_head", "slots", "maskp", "attn")}
            out = (None, (None, None, leaves["attn"]), (leaves["slots_head"], leaves["slots"], leaves["maskp"]))
            crit = TrainLoss(criterion=None, scene_criterion=crit_name, num_action_classes=nb, slot_matching_method="matching",
                             mask_prediction_loss_weight=1.0, mask_distill_loss_weight=3.0, scene_loss_weight=2000)
            total, logits, ld = crit(None, out, (None, base["teacher"].clone()), base["target"], fg_mask=(base["fg"], base["fgN"]))
            total.backward()
            ol = {k: base[k].clone().requires_grad_(True) for k in leaves}
            oout = (None, (None, None, ol["attn"]), (ol["slots_head"], ol["slots"], ol["maskp"]))
            ototal, ologits, old, oidx = ref_cpu.train_loss(cfg, oout, base["teacher"].clone(), base["target"], (base["fg"], base["fgN"]),
                                                            scene_loss_weight=2000, mask_prediction_loss_weight=1.0,
                                                            mask_distill_loss_weight=3.0, scene_criterion=crit_name)
            ototal.backward()
            errs = {"total": rel_err(ototal, total), "logits": rel_err(ologits, logits)}
            errs.update({"d" + k: rel_err(ol[k].grad, leaves[k].grad) for k in leaves})
            errs.update({k: abs(old[k] - ld[k]) / max(abs(ld[k]), 1e-30) for k in ld})
            print(f"[loss_criteria S={S} {crit_name}] oracle vs reference: " + ", ".join(f"{k}={v:.1e}" for k, v in errs.items()), ld)
            assert max(errs.values()) < 2e-6, errs
            pre = f"s{S}.{crit_name}."
            fx[pre + "total"] = np.array(float(total.detach().double()))
            fx[pre + "losses"] = np.array([float(ld[k]) for k in ("action_loss", "scene_loss", "cosine_loss", "mask_prediction_loss", "mask_distill_loss")])
            fx[pre + "logits"] = logits.detach().numpy()
            fx[pre + "match"] = np.stack([oidx[0].numpy(), oidx[1].numpy()], axis=1)
            for k in leaves:
                fx[pre + "d" + k] = leaves[k].grad.numpy()
    path = os.path.join(ROOT, "tests", "golden", "loss_criteria.npz")
    np.savez_compressed(path, **fx)
    print(f"[loss_criteria] wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    ref = install_reference()
    for name in CONFIGS:
        if args.only in (None, name):
            generate(name, ref)
    if args.only in (None, "teacher"):
        generate_teacher(ref)
    if args.only in (None, "keys"):
        generate_state_dict_keys(ref)
    if args.only in (None, "optim"):
        generate_optim(ref)
    if args.only in (None, "fame"):
        generate_fame(ref)
    if args.only in (None, "knn"):
        generate_knn(ref)
    if args.only in (None, "loss"):
        generate_loss_criteria(ref)


if __name__ == "__main__":
    main()
