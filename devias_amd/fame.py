"""FAME on the device: foreground masks from frame differences + an HSV colour model, and foreground/background clip mixing
(reference: utils/transform/fame.py, constructed at run_slot_finetuning.py:422 as FAME(beta=, prob_aug=) and called in the
training step at engine/engine_for_slot.py:106-108 as `samples, targets, masks = mask_model(samples, targets)`).

Same constructor and forward contract as the reference class.  All image arithmetic runs in libdevias_amd.so
(devias_fame_* in include/devias_amd.h); PyTorch allocates the buffers and draws the two random vectors.  The reference
draws `torch.randperm(B)` on the GPU and `torch.rand(B)` on the CPU; both are drawn on the CPU here (no device sync), and
both can be passed in (`index=`, `rand_batch=`) for reproducible tests."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops


def _call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args), name)


class FAME(nn.Module):
    def __init__(self, crop_size=112, beta=0.5, device="cuda", eps=1e-8, prob_aug=0.5):
        super().__init__()
        self.crop_size = crop_size
        self.gauss_size = int(0.1 * crop_size) // 2 * 2 + 1          # fame.py:19-22 (112 -> 11 taps, sigma 11/3, whatever the clip size)
        self.gauss_sigma = self.gauss_size / 3
        self.device, self.eps, self.beta, self.prob_aug = device, eps, beta, prob_aug

    def __repr__(self):                                                # the engine dispatches on `'FAME' in str(mask_model)` (engine_for_slot.py:106)
        return f"FAME(beta={self.beta}, prob_aug={self.prob_aug}, gauss={self.gauss_size})"

    @torch.no_grad()
    def masks(self, videos: torch.Tensor):
        """Binary clip mask uint8 [B,H,W] (fame.py:89-98), its 16x16 pooling [B, hw/256] and the pooled per-frame-pair masks
        [B, T/2, hw/256] (fame.py:100-112, 142-148), in the clips' own order."""
        ops._chk(videos, "FAME.videos", torch.float32)
        B, C, T, H, W = videos.shape
        if C != 3 or T % 2:
            raise ValueError("FAME: clips must be [B, 3, T even, H, W]")
        st, dev, HW, S = ops._stream(), videos.device, H * W, 1 + T // 2
        diffs = torch.empty(B * S, H, W, dtype=torch.float32, device=dev)
        tmp = torch.empty_like(diffs)
        cmap = torch.empty(B, HW, dtype=torch.int16, device=dev)
        _call("devias_fame_diff_color", videos.data_ptr(), B, T, H, W, diffs.data_ptr(), cmap.data_ptr(), st)
        _call("devias_fame_blur", diffs.data_ptr(), tmp.data_ptr(), B * S, H, W, self.gauss_size, self.gauss_sigma, st)
        _call("devias_fame_seg_refine", tmp.data_ptr(), cmap.data_ptr(), B * S, S, HW, self.eps, diffs.data_ptr(), st)
        _call("devias_fame_blur", diffs.data_ptr(), tmp.data_ptr(), B * S, H, W, self.gauss_size, self.gauss_sigma, st)
        num_fg = int(self.beta * HW)
        if num_fg <= 0:                                                # torch.topk(k=0): an empty foreground
            binmask = torch.zeros(B * S, H, W, dtype=torch.uint8, device=dev)
            pooled = torch.zeros(B * S, (H // 16) * (W // 16), dtype=torch.float32, device=dev)
        else:
            binmask = torch.empty(B * S, H, W, dtype=torch.uint8, device=dev)
            pooled = torch.empty(B * S, (H // 16) * (W // 16), dtype=torch.float32, device=dev)
            _call("devias_fame_binarize_pool", tmp.data_ptr(), B * S, H, W, min(num_fg, HW), 16, binmask.data_ptr(), pooled.data_ptr(), st)
        pooled = pooled.view(B, S, -1)
        return binmask.view(B, S, H, W), pooled[:, 0], pooled[:, 1:]

    @torch.no_grad()
    def forward(self, videos, label, center_frame=None, index=None, rand_batch=None):
        B, C, T, H, W = videos.shape
        videos = videos.contiguous()
        binmask, pooled, pooled_pf = self.masks(videos)
        if index is None:
            index = torch.randperm(B)
        if self.prob_aug < 1:
            if rand_batch is None:
                rand_batch = torch.rand(B)
            rb = rand_batch.cpu()
            aug_ind, ori_ind = torch.where(rb < self.prob_aug)[0], torch.where(rb >= self.prob_aug)[0]
            src = torch.cat([aug_ind, ori_ind])
            aug = torch.cat([torch.ones_like(aug_ind), torch.zeros_like(ori_ind)])
        else:
            src, aug = torch.arange(B), torch.ones(B, dtype=torch.int64)
        partner = index.cpu()[src]
        dev = videos.device
        tab = torch.stack([src, partner, aug]).to(torch.int32).to(dev, non_blocking=True)
        out = torch.empty_like(videos)
        S = 1 + T // 2
        _call("devias_fame_mix", videos.data_ptr(), binmask.data_ptr(), S * H * W, tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr(),
              out.data_ptr(), B, C * T, H * W, ops._stream())
        src_d = tab[0].long()
        all_label = label.to(dev)[src_d]
        mask = pooled[src_d].contiguous().to(videos.dtype)
        masks_per_frame = pooled_pf[src_d].reshape(B, -1).to(videos.dtype)
        if center_frame is not None:
            return out, all_label, (mask, masks_per_frame), center_frame.to(dev)[src_d]
        return out, all_label, (mask, masks_per_frame)
