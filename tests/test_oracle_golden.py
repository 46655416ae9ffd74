"""The CPU oracle against the fixtures produced by the REAL reference (tests/golden/make_goldens.py)."""
import numpy as np
import pytest
import torch

from devias_amd import synth
from oracle import ref_cpu

import golden_util as gu

KNOWN_ATTN_MASK_SEED7 = [[0, 1, 0, 1, 0, 1, 1, 0], [1, 1, 0, 0, 1, 0, 0, 0], [0, 1, 0, 1, 0, 1, 1, 0], [1, 0, 1, 0, 0, 0, 1, 1],
                         [0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1, 0, 1], [1, 0, 1, 0, 0, 1, 0, 1], [0, 0, 1, 0, 1, 0, 0, 1]]      # keep 0.5, seed 7, B = H = 1, N = 8


@pytest.mark.parametrize("name", gu.STUDENT_GOLDENS)
def test_oracle_matches_reference_golden(name):
    fx, cfg, B = gu.load(name)
    x, y, tl, fg = gu.inputs(cfg, B)
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    assert list(P.keys()) == [str(n) for n in fx["param_names"]]
    total, logits, ld, grads, out, idx = ref_cpu.train_step(P, cfg, x, y, tl, fg)
    gu.check_against_golden(fx, out, total, logits, ld, grads, tol_out=5e-5, tol_grad=1e-3, idx=idx)


def test_oracle_matches_reference_golden_with_dropout():
    """training mode with drop_rate = attn_drop_rate = drop_path_rate = 0.1: the fixture is the REAL reference run with its nn.Dropout / DropPath modules
    multiplying by given masks (tests/golden/make_goldens.py install_masks); the oracle gets the same masks (golden_util.dropout_masks) -- including the
    attention-matrix mask, which is the numpy restatement of the HIP kernels' hash (ref_cpu.attn_drop_mask)"""
    fx, cfg, B = gu.load(gu.DROPOUT_GOLDEN)
    x, y, tl, fg = gu.inputs(cfg, B)
    P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=0)
    drops = gu.dropout_masks(cfg, B, fx["_rates"])
    total, logits, ld, grads, out, idx = ref_cpu.train_step(P, cfg, x, y, tl, fg, drops=drops)
    gu.check_against_golden(fx, out, total, logits, ld, grads, tol_out=5e-5, tol_grad=1e-3, idx=idx)
    plain, _cfg, _ = gu.load("vitb_t8")
    assert abs(float(fx["total_loss"]) - float(plain["total_loss"])) > 1e-3 * abs(float(plain["total_loss"]))      # it is not the un-dropped step


def test_attn_drop_mask_statistics_and_known_values():
    """the kernels' attention-mask hash, restated in numpy: keep fraction, independence across seeds / heads, and pinned values (a change of the hash
    in csrc/attention.hip without the oracle -- or the reverse -- fails the GPU tests; a change of both fails here)"""
    m = ref_cpu.attn_drop_mask(0.9, 0x0123456789ABCDEF, 2, 3, 128)
    assert m.shape == (2, 3, 128, 128) and abs(float((m > 0).float().mean()) - 0.9) < 0.01
    assert float(m.max()) == float(np.float32(1) / np.float32(0.9))
    m2 = ref_cpu.attn_drop_mask(0.9, 0x0123456789ABCDEF + 1, 2, 3, 128)
    agree = float(((m > 0) == (m2 > 0)).float().mean())
    assert abs(agree - 0.82) < 0.02                                   # independent masks agree with probability 0.81 + 0.01
    assert abs(float(((m[0, 0] > 0) == (m[1, 2] > 0)).float().mean()) - 0.82) < 0.03
    k = (ref_cpu.attn_drop_mask(0.5, 7, 1, 1, 8)[0, 0] > 0).int().numpy()
    assert k.tolist() == KNOWN_ATTN_MASK_SEED7, k.tolist()


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
