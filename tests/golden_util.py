"""Shared helpers for the golden-fixture tests (fixtures are made by tests/golden/make_goldens.py from the real reference)."""
import ast
import os

import numpy as np
import torch

from devias_amd import synth
from oracle import ref_cpu

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STUDENT_GOLDENS = ["vitb_t8", "vitb_t16", "vits_t8", "vitb_t8_s4_untied", "vitb_t8_mlphead"]
DROPOUT_GOLDEN = "vitb_t8_dropout"          # training mode with drop_rate / attn_drop_rate / drop_path_rate = 0.1 and GIVEN masks (dropout_masks below)


def load(name):
    fx = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    kw = ast.literal_eval(str(fx["config"]))
    rates = kw.pop("_drop", None)
    cfg = ref_cpu.SlotViTConfig(**kw)
    if rates is not None:
        fx["_rates"] = rates
    return fx, cfg, int(fx["batch"])


def dropout_masks(cfg, B, rates):
    """the `drops` argument of ref_cpu.student_forward for these rates: formula masks (devias_amd.synth.dropout_mask) for nn.Dropout's element masks and
    drop_path's per-sample masks, and the attention kernels' mask hash (ref_cpu.attn_drop_mask) for attn_drop.  tests/golden/make_goldens.py installs the
    same masks in the reference's own dropout modules."""
    N, D, H = cfg.num_patches, cfg.embed_dim, cfg.num_heads
    kd, ka = 1.0 - rates["drop_rate"], 1.0 - rates["attn_drop_rate"]
    dpr = [x.item() for x in torch.linspace(0, rates["drop_path_rate"], cfg.depth)]
    drops = {}
    if rates["drop_rate"] > 0:
        drops["pos"] = synth.dropout_mask("pos", -1, (B, N, D), kd)
    for i in range(cfg.depth):
        d = {}
        if rates["drop_rate"] > 0:
            d["proj"] = synth.dropout_mask("proj", i, (B, N, D), kd)
            d["mlp"] = synth.dropout_mask("mlp", i, (B, N, D), kd)
        if rates["attn_drop_rate"] > 0:
            d["attn"] = ref_cpu.attn_drop_mask(ka, synth.attn_drop_seed(i), B, H, N)
        if dpr[i] > 0:
            d["path1"] = synth.dropout_mask("path1", i, (B,), 1.0 - dpr[i])
            d["path2"] = synth.dropout_mask("path2", i, (B,), 1.0 - dpr[i])
        drops[i] = d
    return drops


class FormulaDropoutSource:
    """devias_amd.modeling_slot.DropoutSource with the formula masks of dropout_masks(): the HIP model then uses the masks the golden's reference run used"""

    def element_mask(self, kind, block, shape, keep, device):
        return synth.dropout_mask(kind, block, shape, keep).to(device).contiguous()

    def path_scale(self, block, B, keep, device):
        return torch.stack([synth.dropout_mask("path1", block, (B,), keep), synth.dropout_mask("path2", block, (B,), keep)]).to(device).contiguous()

    def attn_seed(self, block):
        return synth.attn_drop_seed(block)


def inputs(cfg, B, seed=1000):
    x = synth.video(B, cfg.all_frames, cfg.img_size, seed=seed)
    y = synth.targets(B, cfg.num_classes, seed=seed)
    tl = synth.teacher_logits(B, cfg.num_scene_classes, seed=seed)
    fg = synth.fg_masks(B, cfg.num_patches, cfg.grid * cfg.grid, seed=seed)
    return x, y, tl, fg


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def sample_idx(name, numel, k=16):
    return (synth.hash_u64(7, "gradsample." + name, k) % np.uint64(numel)).astype(np.int64)


def check_against_golden(fx, out, total, logits, ld, grads, tol_out, tol_grad, idx=None):
    """Compare a (forward outputs, loss, gradients) result with a golden fixture.
    Gradient errors are scaled by max(|g_ref|max, 1e-6 * global max): a handful of gradients are
    mathematically zero (slot-query LayerNorm bias) and hold only round-off in the reference too."""
    (af, sf), (al, sl, attn), (slots_head, slots, maskp) = out
    errs = {
        "slots_head": rel(slots_head.detach().float().cpu(), fx["slots_head"]),
        "slots": rel(slots.detach().float().cpu(), fx["slots"]),
        "mask_predictions": rel(maskp.detach().float().cpu(), fx["mask_predictions"]),
        "attn": rel(attn.detach().float().cpu(), fx["attn"]),
        "action_logit": rel(al.detach().float().cpu(), fx["action_logit"]),
        "scene_logit": rel(sl.detach().float().cpu(), fx["scene_logit"]),
        "action_feat": rel(af.detach().float().cpu(), fx["action_feat"]),
        "scene_feat": rel(sf.detach().float().cpu(), fx["scene_feat"]),
        "matched_logits": rel(logits.detach().float().cpu(), fx["matched_logits"]),
        "total_loss": abs(float(total) - float(fx["total_loss"])) / abs(float(fx["total_loss"])),
    }
    for k, v in zip(fx["loss_names"], fx["loss_values"]):
        errs["loss." + str(k)] = abs(float(ld[str(k)]) - float(v)) / max(abs(float(v)), 1e-30)
    bad = {k: v for k, v in errs.items() if not v <= tol_out}
    assert not bad, f"outputs off golden (tol {tol_out}): {bad}"
    if idx is not None:
        assert list(map(int, idx[0])) == fx["match_action_slot"].tolist()
        assert list(map(int, idx[1])) == fx["match_scene_slot"].tolist()
    gerrs = {}
    if grads is not None:
        names = [str(n) for n in fx["param_names"]]
        nmax = float(fx["grad_norms"].max())
        for i, n in enumerate(names):
            g = grads[n].detach().float().cpu()
            ref_norm = float(fx["grad_norms"][i])
            gerrs[n + ".norm"] = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-6 * nmax)
            s = g.reshape(-1)[torch.from_numpy(sample_idx(n, g.numel()))].double().numpy()
            ref_s = fx["grad_samples"][i].astype(np.float64)
            gerrs[n + ".samples"] = float(np.abs(s - ref_s).max() / max(np.abs(ref_s).max(), 1e-6 * nmax, 1e-30))
        bad = {k: v for k, v in gerrs.items() if not v <= tol_grad}
        assert not bad, f"gradients off golden (tol {tol_grad}): {dict(list(bad.items())[:8])} (+{max(0, len(bad) - 8)} more)"
    return errs, gerrs
