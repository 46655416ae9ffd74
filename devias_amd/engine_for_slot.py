"""Step semantics of engine/engine_for_slot.py for the MI355X path: train_class_batch (:50-56) and a lean
train_one_epoch (:64-214) without per-step host syncs.  The teacher may be a module (forward under no_grad, as in the
reference) or precomputed scene logits (the primary benchmark metric treats them as an input)."""
from __future__ import annotations

import math
import sys
from typing import Iterable, Optional

import torch

from .optim import FusedAdamW


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
    grad_norm = None
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
        if grad_sync is not None:
            # gradient accumulation: only the LAST micro-batch of a window starts the bucket all-reduces (earlier ones just accumulate)
            grad_sync.set_accumulate((data_iter_step + 1) % update_freq != 0)
        loss.backward()
        if (data_iter_step + 1) % update_freq == 0:
            if grad_sync is not None:
                grad_sync.finish()
            if isinstance(optimizer, FusedAdamW) and optimizer.multi_tensor:
                # gradient norm + clip_grad_norm_ (utils/utils.py:388-394) fused into the update; the norm stays on the device
                optimizer.step(max_norm=float(max_norm or 0.0))
                grad_norm = optimizer.last_grad_norm
            else:
                if max_norm and max_norm > 0:
                    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
                optimizer.step()
            optimizer.zero_grad(set_to_none=True)
        n_steps += 1
        if check_finite_every and n_steps % check_finite_every == 0:
            loss_value = float(loss.detach().float().sum())
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            stats["loss"] = loss_value
            if grad_norm is not None:
                stats["grad_norm"] = float(grad_norm)
            stats["lr"] = max(g["lr"] for g in optimizer.param_groups)
            stats["min_lr"] = min(g["lr"] for g in optimizer.param_groups)
            stats.update({k: float(v) for k, v in loss_dict.items()})
    return stats


@torch.no_grad()
def validation_one_epoch(data_loader, model, device, topk=(1, 5)):
    """engine/engine_for_slot.py:215-250: eval-mode forward, cross-entropy of the selected action logits [B, nb + ns] against
    the action target, top-1 / top-5 accuracy.  Sums are kept on the device; one host read at the end of the loader (the
    reference reads three scalars per batch).  Returns {'loss', 'acc1', 'acc5'} as sample-weighted means (acc in percent)."""
    model.eval()
    acc = torch.zeros(2 + len(topk), dtype=torch.float64, device=device)          # [n, sum of CE, hits@k...]
    for batch in data_loader:
        videos = batch[0].to(device, non_blocking=True)
        target = batch[1].to(device, non_blocking=True)
        _, (output, _scene_output, _attn), _ = model(videos)
        acc += _batch_metrics(output.float(), target, topk)
    n, ce, hits = float(acc[0]), float(acc[1]), [float(v) for v in acc[2:]]
    out = {"loss": ce / max(n, 1.0)}
    for k, h in zip(topk, hits):
        out["acc%d" % k] = 100.0 * h / max(n, 1.0)
    return out


def _batch_metrics(output, target, topk):
    """[B, sum of per-sample CE, top-k hit counts] of one batch (timm.utils.accuracy semantics: a hit if the target is among the
    k largest logits).  [B, 765] metric arithmetic, not part of the hot path."""
    logp = torch.log_softmax(output, dim=-1)
    ce = -logp.gather(1, target.view(-1, 1)).sum()
    _, pred = output.topk(max(topk), dim=1)
    hit = pred.eq(target.view(-1, 1))
    vals = [torch.tensor(float(output.shape[0]), device=output.device, dtype=torch.float64), ce.double()]
    vals += [hit[:, :k].any(dim=1).sum().double() for k in topk]
    return torch.stack(vals)


@torch.no_grad()
def final_test(data_loader, model, device, file):
    """engine/engine_for_slot.py:253-303: as validation, and one line `id [logits] target chunk split` per sample written to
    `file` after a first line with the LAST batch's `acc1, acc5` (the reference writes exactly that)."""
    model.eval()
    acc = torch.zeros(4, dtype=torch.float64, device=device)
    lines, last = [], (0.0, 0.0)
    for batch in data_loader:
        videos = batch[0].to(device, non_blocking=True)
        target = batch[1].to(device, non_blocking=True)
        ids, chunk_nb, split_nb = batch[2], batch[3], batch[4]
        _, (output, _scene_output, _attn), _ = model(videos)
        m = _batch_metrics(output.float(), target, (1, 5))
        acc += m
        rows, tgt, mh = output.float().cpu().numpy(), target.cpu().numpy(), m.cpu().numpy()
        last = (100.0 * mh[2] / mh[0], 100.0 * mh[3] / mh[0])
        for i in range(rows.shape[0]):
            lines.append("{} {} {} {} {}\n".format(ids[i], str(rows[i].tolist()), str(int(tgt[i])), str(int(chunk_nb[i])), str(int(split_nb[i]))))
    with open(file, "w") as f:
        f.write("{}, {}\n".format(last[0], last[1]))
        f.writelines(lines)
    n = max(float(acc[0]), 1.0)
    return {"loss": float(acc[1]) / n, "acc1": 100.0 * float(acc[2]) / n, "acc5": 100.0 * float(acc[3]) / n}
