"""Parameter groups, layer-wise LR decay and the LR / weight-decay schedules of the reference, for the fused optimizer.

Mirrors the surface `run_slot_finetuning.py:529-560` drives:
  get_num_layer_for_vit      utils/optim_factory.py:24-35
  LayerDecayValueAssigner    utils/optim_factory.py:38-46
  get_parameter_groups       utils/optim_factory.py:49-93   (layer_<id>_{decay,no_decay} groups; every agg_block group: scale 0.1)
  create_optimizer           utils/optim_factory.py:96-133  (opt='adamw' -> devias_amd.optim.FusedAdamW; the reference's other
                                                             optimizers are timm/apex classes outside the hot path)
  cosine_scheduler           utils/utils.py:424-441
Pure host logic; pinned against the reference by tests/golden/optim_factory.json (tests/golden/make_goldens.py)."""
from __future__ import annotations

import json
import math

import numpy as np


_STEM_NAMES = ("cls_token", "mask_token", "pos_embed")


def get_num_layer_for_vit(var_name: str, num_max_layer: int) -> int:
    """Depth index of a parameter for layer-wise LR decay: stem (tokens, pos_embed, patch_embed) -> 0, `blocks.<i>.` -> i + 1,
    everything after the encoder (norm, agg_block, heads, rel_pos_bias) -> num_max_layer - 1."""
    if var_name in _STEM_NAMES or var_name.startswith("patch_embed"):
        return 0
    if var_name.startswith("blocks"):
        return 1 + int(var_name.split(".")[1])
    return num_max_layer - 1


class LayerDecayValueAssigner:
    """values[i] = layer_decay ** (num_layers + 1 - i), i in [0, num_layers + 1] (run_slot_finetuning.py:531-534)."""

    def __init__(self, values):
        self.values = list(values)

    @classmethod
    def from_decay(cls, layer_decay: float, num_layers: int):
        return cls([layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)])

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


def _group_key(name: str, decays: bool, layer_id):
    key = "decay" if decays else "no_decay"
    if layer_id is not None:
        key = f"layer_{layer_id}_{key}"
        if "agg_block" in name:
            key = "agg_block_" + key
    return key


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None, agg_block_scale=0.1,
                         verbose=False, return_names=False):
    """Optimizer groups in first-seen order of model.named_parameters(): 1-D tensors, `.bias` and skip_list names do not decay;
    with `get_num_layer` the groups split per depth index, and agg_block parameters get groups of their own.  Two quirks of the
    reference are kept because they decide the learning rates: a group's lr_scale is fixed by the FIRST parameter that opens
    it, and a group opened by an agg_block parameter takes `agg_block_scale` whatever its depth index says."""
    table = {}                                   # key -> (weight_decay, lr_scale, [params], [names]); dicts keep insertion order
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        decays = not (param.dim() == 1 or name.endswith(".bias") or name in skip_list)
        layer_id = get_num_layer(name) if get_num_layer is not None else None
        key = _group_key(name, decays, layer_id)
        entry = table.get(key)
        if entry is None:
            if "agg_block" in name:
                lr_scale = agg_block_scale
            else:
                lr_scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.0
            entry = table[key] = (weight_decay if decays else 0.0, lr_scale, [], [])
        entry[2].append(param)
        entry[3].append(name)
    groups = [{"weight_decay": wd, "params": ps, "lr_scale": sc} for wd, sc, ps, _ in table.values()]
    names = {k: {"weight_decay": wd, "params": ns, "lr_scale": sc} for k, (wd, sc, _, ns) in table.items()}
    if verbose:
        print("Param groups = %s" % json.dumps(names, indent=2))
    return (groups, names) if return_names else groups


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """`args` needs .opt, .lr, .weight_decay and optionally .opt_eps / .opt_betas (run_slot_finetuning.py:64-76).  With a
    non-zero weight decay the decay moves into the groups and the optimizer default becomes 0, as in the reference."""
    from .optim import FusedAdamW

    if args.opt.lower().split("_")[-1] != "adamw":
        raise ValueError(f"devias_amd implements the reference's default optimizer (adamw) on the HIP path; got opt={args.opt!r}")
    hyper = {"lr": args.lr, "weight_decay": args.weight_decay}
    if args.weight_decay and filter_bias_and_bn:
        if skip_list is None:
            skip_list = model.no_weight_decay() if hasattr(model, "no_weight_decay") else ()
        params = get_parameter_groups(model, args.weight_decay, skip_list, get_num_layer, get_layer_scale)
        hyper["weight_decay"] = 0.0
    else:
        params = model.parameters()
    if getattr(args, "opt_eps", None) is not None:
        hyper["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        hyper["betas"] = tuple(args.opt_betas)
    return FusedAdamW(params, **hyper)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Per-iteration schedule of length epochs * niter_per_ep: linear warm-up from start_warmup_value to base_value, then
    half-cosine to final_value.  As in the reference, `warmup_steps` > 0 overrides the warm-up LENGTH but the ramp is only
    emitted when warmup_epochs > 0."""
    total = epochs * niter_per_ep
    n_warm = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    ramp = np.linspace(start_warmup_value, base_value, n_warm) if warmup_epochs > 0 else np.zeros(0)
    n_cos = total - n_warm
    phase = np.array([math.cos(math.pi * i / n_cos) for i in range(n_cos)])
    schedule = np.concatenate((ramp, final_value + 0.5 * (base_value - final_value) * (1 + phase)))
    assert len(schedule) == total
    return schedule
