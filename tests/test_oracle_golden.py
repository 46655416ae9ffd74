"""The CPU oracle against the fixtures produced by the REAL reference (tests/golden/make_goldens.py)."""
import numpy as np
import pytest
import torch

from devias_amd import synth
from oracle import ref_cpu

import golden_util as gu


@pytest.mark.parametrize("name", gu.STUDENT_GOLDENS)
def test_oracle_matches_reference_golden(name):
    fx, cfg, B = gu.load(name)
    x, y, tl, fg = gu.inputs(cfg, B)
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    assert list(P.keys()) == [str(n) for n in fx["param_names"]]
    total, logits, ld, grads, out, idx = ref_cpu.train_step(P, cfg, x, y, tl, fg)
    gu.check_against_golden(fx, out, total, logits, ld, grads, tol_out=5e-5, tol_grad=1e-3, idx=idx)


def test_oracle_teacher_matches_reference_golden():
    fx = dict(np.load(gu.GOLDEN_DIR + "/teacher_vitb_t8.npz"))
    cfg = ref_cpu.SlotViTConfig(all_frames=8)
    P = synth.fill_params(ref_cpu.teacher_param_shapes(cfg), seed=1)
    with torch.no_grad():
        tok, logits = ref_cpu.teacher_forward(P, cfg, synth.video(2, 8, 224, seed=1000))
    assert gu.rel(tok, fx["token"]) < 5e-5 and gu.rel(logits, fx["logits"]) < 5e-5


def test_match_equals_scipy_hungarian():
    """brute-force ordered-pair argmin == scipy.optimize.linear_sum_assignment on S x 2 costs
    (reference utils/loss/train_loss.py:121)."""
    from scipy.optimize import linear_sum_assignment
    g = torch.Generator().manual_seed(3)
    for S in (2, 3, 4):
        for _ in range(200):
            cost = -torch.rand(S, 2, generator=g)
            r, c = linear_sum_assignment(cost.numpy())
            i = int(r[list(c).index(0)]); j = int(r[list(c).index(1)])
            assert ref_cpu.match_slots(cost) == (i, j)


def test_sinusoid_table_known_values():
    t = ref_cpu.sinusoid_table(4, 8)[0]
    assert t.shape == (4, 8)
    assert torch.allclose(t[0], torch.tensor([0., 1., 0., 1., 0., 1., 0., 1.]))
    assert abs(float(t[1, 0]) - np.sin(1.0)) < 1e-7 and abs(float(t[1, 1]) - np.cos(1.0)) < 1e-7
    assert abs(float(t[3, 2]) - np.sin(3.0 / 10000 ** (2 / 8))) < 1e-7


@pytest.mark.parametrize("S", [2, 3])
@pytest.mark.parametrize("crit", ["KL", "CE"])
def test_oracle_loss_criteria_match_reference_golden(S, crit):
    """oracle train_loss, both scene criteria, against the reference's own TrainLoss on the committed inputs
    (tests/golden/loss_criteria.npz, made by make_goldens.py --only loss)."""
    fx = dict(np.load(gu.GOLDEN_DIR + "/loss_criteria.npz"))
    t = {k: torch.from_numpy(fx[f"s{S}.{k}"]) for k in ("slots_head", "slots", "maskp", "attn", "teacher", "target", "fg", "fgN")}
    lv = {k: t[k].clone().requires_grad_(True) for k in ("slots_head", "slots", "maskp", "attn")}
    cfg = ref_cpu.SlotViTConfig(all_frames=8, num_latents=S)
    out = (None, (None, None, lv["attn"]), (lv["slots_head"], lv["slots"], lv["maskp"]))
    total, logits, ld, idx = ref_cpu.train_loss(cfg, out, t["teacher"], t["target"], (t["fg"], t["fgN"]), scene_loss_weight=2000,
                                                mask_prediction_loss_weight=1.0, mask_distill_loss_weight=3.0, scene_criterion=crit)
    total.backward()
    pre = f"s{S}.{crit}."
    assert abs(float(total.detach()) - float(fx[pre + "total"])) < 1e-5 * abs(float(fx[pre + "total"]))
    got = [ld[k] for k in ("action_loss", "scene_loss", "cosine_loss", "mask_prediction_loss", "mask_distill_loss")]
    assert np.allclose(got, fx[pre + "losses"], rtol=1e-5, atol=1e-7)
    assert torch.stack(idx, dim=1).tolist() == fx[pre + "match"].tolist()
    assert gu.rel(logits.detach(), fx[pre + "logits"]) < 1e-6
    for k in lv:
        assert gu.rel(lv[k].grad, fx[pre + "d" + k]) < 1e-5, k
