"""Step semantics of engine/engine_for_slot.py for the MI355X path: train_class_batch (:50-56) and a lean
train_one_epoch (:64-214) without per-step host syncs.  The teacher may be a module (forward under no_grad, as in the
reference) or precomputed scene logits (the primary benchmark metric treats them as an input)."""
from __future__ import annotations

import math
import sys
from typing import Iterable, Optional

import torch


def train_class_batch(model, scene_model, samples, target, train_criterion, fg_mask=None):
    """engine/engine_for_slot.py:50-56.  `scene_model` is a module returning (token, logits) or a [B, 365] logits tensor."""
    student_output = model(samples)
    if torch.is_tensor(scene_model):
        teacher_output = (None, scene_model)
    else:
        with torch.no_grad():
            teacher_output = scene_model(samples, return_attn=False)
    total_loss, output, loss_dict = train_criterion(model, student_output, teacher_output, target, fg_mask=fg_mask)
    return total_loss, output, loss_dict


def train_one_epoch(model, scene_model, train_criterion, data_loader: Iterable, optimizer, device, epoch: int,
                    max_norm: float = 0, start_steps: int = 0, lr_schedule_values=None, wd_schedule_values=None,
                    num_training_steps_per_epoch: Optional[int] = None, update_freq: int = 1, mask_model=None,
                    grad_sync=None, check_finite_every: int = 50, log_every: int = 100):
    """engine/engine_for_slot.py:64-214 restated for this stack: LR/WD schedule poke (:91-96), H2D (:98-99), mask model
    (:106-108), train_class_batch, backward, optional gradient all-reduce (`grad_sync`, devias_amd.parallel), optimizer step.
    The per-step `loss.item()` finite check (:140-144) and `torch.cuda.synchronize()` (:171) are replaced by one
    host check every `check_finite_every` steps."""
    model.train(True)
    if not torch.is_tensor(scene_model):
        scene_model.eval()
    optimizer.zero_grad(set_to_none=True)
    stats = {}
    n_steps = 0
    for data_iter_step, batch in enumerate(data_loader):
        samples, targets = batch[0], batch[1]
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if (lr_schedule_values is not None or wd_schedule_values is not None) and data_iter_step % update_freq == 0:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group.get("lr_scale", 1.0)
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if mask_model is not None:
            samples, targets, masks = mask_model(samples, targets)
        else:
            masks = batch[2]
            masks = tuple(m.to(device, non_blocking=True) for m in masks)
        teacher = scene_model if not torch.is_tensor(scene_model) else scene_model
        loss, output, loss_dict = train_class_batch(model, teacher, samples, targets, train_criterion, fg_mask=masks)
        if update_freq > 1:
            loss = loss / update_freq
        loss.backward()
        if (data_iter_step + 1) % update_freq == 0:
            if grad_sync is not None:
                grad_sync.finish()
            if max_norm and max_norm > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
        n_steps += 1
        if check_finite_every and n_steps % check_finite_every == 0:
            loss_value = float(loss.detach().float().sum())
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            stats["loss"] = loss_value
            stats.update({k: float(v) for k, v in loss_dict.items()})
    return stats
