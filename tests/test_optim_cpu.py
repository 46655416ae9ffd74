"""Host logic of the optimizer surface against values produced by the REAL reference (tests/golden/optim_factory.json,
written by tests/golden/make_goldens.py --only optim): parameter groups with layer decay, cosine schedules."""
import json
import os
import types

import pytest

import devias_amd
from devias_amd import optim_factory as of

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "optim_factory.json")))
CTOR = {
    "tied_s2_d8": dict(num_classes=400, all_frames=8, num_latents=2, slot_matching_method="matching", agg_weights_tie=True, agg_depth=8,
                       num_scene_classes=365),
    "untied_s4_d4": dict(num_classes=400, all_frames=8, num_latents=4, slot_matching_method="matching", agg_weights_tie=False, agg_depth=4,
                         num_scene_classes=365),
}
_models = {}


def _model(name):
    if name not in _models:
        _models[name] = devias_amd.create_model("slot_vit_base_patch16_224", **CTOR[name])
    return _models[name]


@pytest.mark.parametrize("case", sorted(GOLD["groups"]))
def test_parameter_groups_match_reference(case):
    g = GOLD["groups"][case]
    m = _model(case.split(".")[0])
    assert m.get_num_layers() == g["num_layers"]
    assigner = of.LayerDecayValueAssigner.from_decay(g["layer_decay"], g["num_layers"]) if g["layer_decay"] < 1.0 else None
    groups, names = of.get_parameter_groups(m, g["weight_decay"], m.no_weight_decay(), assigner.get_layer_id if assigner else None,
                                            assigner.get_scale if assigner else None, agg_block_scale=g["agg_block_scale"], return_names=True)
    got = [[k, v["weight_decay"], v["lr_scale"], v["params"]] for k, v in names.items()]
    assert [x[0] for x in got] == [x[0] for x in g["groups"]]              # same groups, same order
    for a, b in zip(got, g["groups"]):
        assert a[1] == b[1] and a[2] == b[2], (a[0], a[1:3], b[1:3])      # weight decay and lr scale: exact
        assert a[3] == b[3], a[0]                                          # same parameter names in the same order
    by_name = dict(m.named_parameters())
    for grp, (_, _, _, pnames) in zip(groups, got):
        assert [id(p) for p in grp["params"]] == [id(by_name[n]) for n in pnames]


def test_layer_ids():
    assert of.get_num_layer_for_vit("pos_embed", 14) == 0
    assert of.get_num_layer_for_vit("patch_embed.proj.weight", 14) == 0
    assert of.get_num_layer_for_vit("blocks.11.mlp.fc2.bias", 14) == 12
    assert of.get_num_layer_for_vit("agg_block.latents", 14) == 13
    assert of.get_num_layer_for_vit("head.weight", 14) == 13


@pytest.mark.parametrize("i", range(len(GOLD["schedules"])))
def test_cosine_scheduler_bit_exact(i):
    s = GOLD["schedules"][i]
    if s["values"] == "AssertionError":
        with pytest.raises(AssertionError):
            of.cosine_scheduler(**s["kwargs"])
        return
    got = of.cosine_scheduler(**s["kwargs"])
    assert [float.hex(float(v)) for v in got] == s["values"]


def test_create_optimizer_groups_and_errors():
    m = _model("tied_s2_d8")
    assigner = of.LayerDecayValueAssigner.from_decay(0.75, m.get_num_layers())
    args = types.SimpleNamespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    opt = of.create_optimizer(args, m, get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)
    assert len(opt.param_groups) == 30
    assert all("lr_scale" in g for g in opt.param_groups)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(list(m.parameters()))
    assert opt.defaults["weight_decay"] == 0.0 and opt.defaults["betas"] == (0.9, 0.999)
    with pytest.raises(ValueError):
        of.create_optimizer(types.SimpleNamespace(opt="sgd", lr=0.1, weight_decay=0.0), m)
