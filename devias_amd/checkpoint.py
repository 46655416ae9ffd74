"""Checkpoint formats on either side of the slot-ViT path, restated from the reference so that published
VideoMAE / DEVIAS `.pth` files load into the MI355X modules and what is saved here loads back into the reference:

  * fine-tune load: run_slot_finetuning.py:438-499  (top-level 'model' | 'module' key, 'backbone.' / 'encoder.' prefix strip,
    head dropped on shape mismatch, bicubic interpolation of a learnable pos_embed, then utils.load_state_dict)
  * utils.load_state_dict: utils/utils.py:330-375 (non-strict recursive _load_from_state_dict with reporting)
  * save / auto-resume: utils/utils.py:442-517 ('model','optimizer','epoch','scaler','args'[, 'model_ema'] -> checkpoint-<epoch>.pth)

Pure host-side I/O: no kernels involved; the state_dict key/shape contract is what makes it work (tests/test_checkpoint_cpu.py
checks it against key lists dumped from the real reference modules)."""
from __future__ import annotations

import glob
import os
from collections import OrderedDict
from typing import Optional

import torch


def select_model_state(checkpoint: dict, model_key: str = "model|module") -> dict:
    for k in model_key.split("|"):
        if k in checkpoint:
            return checkpoint[k]
    return checkpoint


def prepare_finetune_state_dict(model: torch.nn.Module, checkpoint: dict, num_frames: int, model_key: str = "model|module") -> OrderedDict:
    """run_slot_finetuning.py:447-497"""
    ck = dict(select_model_state(checkpoint, model_key))
    own = model.state_dict()
    for k in ("head.weight", "head.bias"):
        if k in ck and k in own and ck[k].shape != own[k].shape:
            print(f"Removing key {k} from pretrained checkpoint")
            del ck[k]
    new = OrderedDict()
    for key, v in ck.items():
        if key.startswith("backbone."):
            new[key[9:]] = v
        elif key.startswith("encoder."):
            new[key[8:]] = v
        else:
            new[key] = v
    if "pos_embed" in new:
        pe = new["pos_embed"]
        emb = pe.shape[-1]
        num_patches = model.patch_embed.num_patches
        num_extra = model.pos_embed.shape[-2] - num_patches
        t = num_frames // model.patch_embed.tubelet_size
        orig = int(((pe.shape[-2] - num_extra) // t) ** 0.5)
        newsz = int((num_patches // t) ** 0.5)
        if orig != newsz:
            print("Position interpolate from %dx%d to %dx%d" % (orig, orig, newsz, newsz))
            extra, pos = pe[:, :num_extra], pe[:, num_extra:]
            pos = pos.reshape(-1, t, orig, orig, emb).reshape(-1, orig, orig, emb).permute(0, 3, 1, 2)
            pos = torch.nn.functional.interpolate(pos, size=(newsz, newsz), mode="bicubic", align_corners=False)
            pos = pos.permute(0, 2, 3, 1).reshape(-1, t, newsz, newsz, emb).flatten(1, 3)
            new["pos_embed"] = torch.cat((extra, pos), dim=1)
    return new


def load_state_dict(model: torch.nn.Module, state_dict: dict, prefix: str = "", ignore_missing: str = "relative_position_index"):
    """utils/utils.py:330-375.  Returns (missing, unexpected, errors) besides printing like the reference."""
    missing, unexpected, errors = [], [], []
    metadata = getattr(state_dict, "_metadata", None)
    state_dict = state_dict.copy()
    if metadata is not None:
        state_dict._metadata = metadata

    def load(module, pfx=""):
        local = {} if metadata is None else metadata.get(pfx[:-1], {})
        module._load_from_state_dict(state_dict, pfx, local, True, missing, unexpected, errors)
        for name, child in module._modules.items():
            if child is not None:
                load(child, pfx + name + ".")

    load(model, prefix)
    from .modeling_slot import invalidate_weight_cache
    invalidate_weight_cache()          # belt and braces: copy_ under no_grad already moves Tensor._version
    warn = [k for k in missing if not any(ig in k for ig in ignore_missing.split("|"))]
    if warn:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, warn))
    if unexpected:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, unexpected))
    if errors:
        print("\n".join(errors))
    return warn, unexpected, errors


def save_checkpoint(output_dir: str, epoch, model: torch.nn.Module, optimizer=None, scaler=None, args=None, model_ema=None) -> str:
    """utils/utils.py:442-464 (torch.amp branch): checkpoint-<epoch>.pth with the reference's keys."""
    os.makedirs(output_dir, exist_ok=True)
    to_save = {"model": model.state_dict(), "epoch": epoch, "args": args}
    if optimizer is not None:
        to_save["optimizer"] = optimizer.state_dict()
    if scaler is not None:
        to_save["scaler"] = scaler.state_dict()
    if model_ema is not None:
        to_save["model_ema"] = model_ema.state_dict()
    path = os.path.join(output_dir, "checkpoint-%s.pth" % str(epoch))
    torch.save(to_save, path)
    return path


def find_latest_checkpoint(output_dir: str) -> Optional[str]:
    """utils/utils.py:471-481"""
    latest = -1
    for ck in glob.glob(os.path.join(output_dir, "checkpoint-*.pth")):
        t = ck.split("-")[-1].split(".")[0]
        if t.isdigit():
            latest = max(int(t), latest)
    return os.path.join(output_dir, "checkpoint-%d.pth" % latest) if latest >= 0 else None


def auto_resume(output_dir: str, model: torch.nn.Module, optimizer=None, scaler=None) -> int:
    """utils/utils.py:467-500: returns the epoch to start from (0 when nothing to resume)."""
    path = find_latest_checkpoint(output_dir)
    if path is None:
        return 0
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model"])
    print("Resume checkpoint %s" % path)
    start = 0
    if "optimizer" in ck and "epoch" in ck:
        if optimizer is not None:
            optimizer.load_state_dict(ck["optimizer"])
        start = ck["epoch"] + 1
        if scaler is not None and "scaler" in ck:
            scaler.load_state_dict(ck["scaler"])
    return start
