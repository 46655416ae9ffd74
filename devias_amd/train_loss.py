"""TrainLoss with the reference's interface (utils/loss/train_loss.py:7-187), computed by ONE fused HIP launch
(devias_head_match_loss_fwd) instead of B host-side SciPy assignments and six .item() syncs."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops

LOSS_NAMES = ("action_loss", "scene_loss", "cosine_loss", "mask_prediction_loss", "mask_distill_loss")


class HeadMatchLossFn(Function):
    @staticmethod
    def forward(ctx, slots_head, slots, maskp, attn, teacher, target, fg, fgN, nb, w_scene, w_mp, w_md, scene_ce=False):
        slots_head, slots, maskp, attn = (t.contiguous() for t in (slots_head, slots, maskp, attn))
        losses, match, logits = ops.head_match_loss_fwd(slots_head, slots, maskp, attn, teacher, target, fg, fgN, nb, w_scene, w_mp, w_md, scene_ce)
        ctx.saved = (slots_head, slots, maskp, attn, teacher, target, fg, fgN, match)
        ctx.w = (nb, w_scene, w_mp, w_md, scene_ce)
        ctx.mark_non_differentiable(losses, match, logits)
        total = losses[5:6].clone()
        return total, losses, match, logits

    @staticmethod
    def backward(ctx, g_total, *_):
        slots_head, slots, maskp, attn, teacher, target, fg, fgN, match = ctx.saved
        nb, w_scene, w_mp, w_md, scene_ce = ctx.w
        g = g_total.reshape(1).float().contiguous()
        dZ, dslots, dmask, dattn = ops.head_match_loss_bwd(slots_head, slots, maskp, attn, teacher, target, fg, fgN, match, g,
                                                           nb, w_scene, w_mp, w_md, scene_ce)
        return dZ, dslots, dmask, dattn, None, None, None, None, None, None, None, None, None


class TrainLoss(nn.Module):
    """Drop-in for utils.loss.train_loss.TrainLoss ('matching'; scene_criterion 'KL' or 'CE').

    forward(model, student_output, teacher_outputs, target, fg_mask) -> (total_loss[1], action_logit[B,C], loss_dict)
    `loss_dict` holds Python floats like the reference (MetricLogger asserts float, utils/utils.py:94), which costs ONE
    device->host copy of 6 floats; pass sync_loss_dict=False to get 0-d device tensors instead (no host sync at all).
    """

    def __init__(self, criterion=None, scene_criterion="KL", num_action_classes: int = 400, slot_matching_method="matching",
                 scene_loss_weight=2000, mask_prediction_loss_weight=1, mask_distill_loss_weight=3, sync_loss_dict=True):
        super().__init__()
        if slot_matching_method != "matching":
            raise NotImplementedError("only the 'matching' branch is reachable with the slot model (SURVEY.md §2.1 #5)")
        if scene_criterion not in ("KL", "CE"):                 # the reference silently adds no scene term for anything else (:155-158)
            raise ValueError(f"scene_criterion must be 'KL' or 'CE' (run_slot_finetuning.py:57), got {scene_criterion!r}")
        self.criterion = criterion            # accepted, never used in the matching branch (as in the reference)
        self.scene_criterion = scene_criterion
        self.num_action_classes = num_action_classes
        self.num_scene_classes = 365
        self.slot_matching_method = slot_matching_method
        self.mask_prediction_loss_weight = float(mask_prediction_loss_weight)
        self.mask_distill_loss_weight = float(mask_distill_loss_weight)
        self.scene_loss_weight = float(scene_loss_weight)
        self.sync_loss_dict = sync_loss_dict
        self.last_match = None

    def forward(self, model, student_output, teacher_outputs, target, fg_mask=None):
        _, (_, _, attn), (slots_head, slots, mask_predictions) = student_output
        _, teacher_scene_logit = teacher_outputs
        fg, fgN = fg_mask
        dev = slots_head.device
        teacher = teacher_scene_logit.detach().to(device=dev, dtype=torch.float32).contiguous()
        fg = fg.to(device=dev, dtype=torch.float32).contiguous()       # k/256 masks: the reference's .half() is value-preserving
        fgN = fgN.to(device=dev, dtype=torch.float32).contiguous()
        target = target.to(device=dev, dtype=torch.int64).contiguous()
        total, losses, match, logits = HeadMatchLossFn.apply(
            slots_head, slots, mask_predictions, attn, teacher, target, fg, fgN, self.num_action_classes,
            self.scene_loss_weight, self.mask_prediction_loss_weight, self.mask_distill_loss_weight, self.scene_criterion == "CE")
        self.last_match = match
        if self.sync_loss_dict:
            vals = losses.tolist()
            loss_dict = {k: vals[i] for i, k in enumerate(LOSS_NAMES)}
        else:
            loss_dict = {k: losses[i] for i, k in enumerate(LOSS_NAMES)}
        return total, logits, loss_dict
