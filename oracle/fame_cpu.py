"""CPU oracle of the FAME foreground-mask / clip-mixing step (TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_goldens.py
and nothing on the product path).

Restates utils/transform/fame.py of the reference (functions cite file:line).  Two of its operations live in a third-party
dependency that is neither vendored nor version-pinned by the reference (`kornia`, docs/INSTALL.md:32, absent from this
image): `kornia.filters.GaussianBlur2d` and `kornia.color.rgb_to_hsv`.  They are restated here from kornia's published
algorithm (0.6/0.7 line: separable normalised Gaussian taps exp(-x^2/(2 sigma^2)), 'reflect' border; HSV with hue in
[0, 2 pi]) -- PARITY UNPINNED for those two functions.  Everything else is pinned: tests/golden/make_goldens.py runs the reference's
own FAME class with these two restatements injected as the `kornia` module and checks this file against it; the outputs are
committed as tests/golden/fame_*.npz."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

FRAME_MEAN = (0.485, 0.456, 0.406)
FRAME_STD = (0.229, 0.224, 0.225)


# ---- kornia restatements (parity unpinned) ---------------------------------------------------------------------------
def gaussian_kernel1d(ksize: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    if ksize % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2.0 * sigma * sigma))
    return g / g.sum()


def gaussian_blur2d(x: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """kornia.filters.gaussian_blur2d(x[B,C,H,W], (k,k), (s,s), border_type='reflect', separable=True)."""
    k = gaussian_kernel1d(ksize, sigma).to(x.dtype)
    pad = ksize // 2
    B, C, H, W = x.shape
    xp = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    xp = F.conv2d(xp.reshape(B * C, 1, H + 2 * pad, W + 2 * pad), k.view(1, 1, 1, ksize))     # along x
    xp = F.conv2d(xp, k.view(1, 1, ksize, 1))                                                 # along y
    return xp.reshape(B, C, H, W)


def rgb_to_hsv(image: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """kornia.color.rgb_to_hsv: image[..., 3, H, W] in [0, 1] -> (h in [0, 2 pi], s, v)."""
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb = image.min(-3)[0]
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    h = torch.stack((h1, h2, h3), dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = 2.0 * math.pi * h
    return torch.stack((h, s, v), dim=-3)


# ---- FAME (utils/transform/fame.py) -------------------------------------------------------------------------------------
class FameOracle:
    def __init__(self, crop_size=112, beta=0.5, eps=1e-8, prob_aug=0.5):
        """fame.py:14-27 (the driver constructs FAME(beta=, prob_aug=) -> crop_size stays 112: an 11-tap blur, sigma 11/3)."""
        self.ksize = int(0.1 * crop_size) // 2 * 2 + 1
        self.sigma = self.ksize / 3
        self.eps, self.beta, self.prob_aug = eps, beta, prob_aug

    def gauss(self, x):
        return gaussian_blur2d(x, self.ksize, self.sigma)

    def norm_batch(self, m):
        """fame.py:30-36: subtract the per-image minimum, divide by (maximum of the result + eps)."""
        B, H, W = m.shape
        m = m.flatten(1)
        m = m - m.min(dim=-1, keepdim=True)[0]
        m = m / (m.max(dim=-1, keepdim=True)[0] + self.eps)
        return m.reshape(B, H, W)

    def color_map(self, clips):
        """fame.py:48-64: HSV of the temporal mean image -> bin index in [0, 1000] per pixel (10 x 10 x 10 bins, 1-based)."""
        B, C, T, H, W = clips.shape
        hsv = rgb_to_hsv(clips.mean(dim=2))
        img_h, img_s, img_v = hsv[:, 0], hsv[:, 1], hsv[:, 2]
        hx = (img_s * torch.cos(img_h * 2 * math.pi) + 1) / 2
        hy = (img_s * torch.sin(img_h * 2 * math.pi) + 1) / 2
        h = torch.round(hx * 9 + 1)
        s = torch.round(hy * 9 + 1)
        v = torch.round(img_v * 9 + 1)
        return (h + (s - 1) * 10 + (v - 1) * 100).reshape(B, -1).long()

    def get_seg(self, mask, clips):
        """fame.py:44-87."""
        B, C, T, H, W = clips.shape
        cmap = self.color_map(clips)
        fg_idx = torch.topk(mask.reshape(B, -1), k=int(0.5 * H * W), dim=-1)[1]
        bg_idx = torch.topk(mask.reshape(B, -1), k=int(0.1 * H * W), dim=-1, largest=False)[1]
        col_fg, col_bg = cmap.gather(1, fg_idx), cmap.gather(1, bg_idx)
        dict_fg = torch.zeros(B, 1000, dtype=torch.long).scatter_add_(1, col_fg, torch.ones_like(col_fg)).float()
        dict_bg = torch.zeros(B, 1000, dtype=torch.long).scatter_add_(1, col_bg, torch.ones_like(col_bg)).float() + 1
        dict_fg = dict_fg / (dict_fg.sum(-1, keepdim=True) + self.eps)
        dict_bg = dict_bg / (dict_bg.sum(-1, keepdim=True) + self.eps)
        pr_fg, pr_bg = dict_fg.gather(1, cmap), dict_bg.gather(1, cmap)
        refine = pr_fg / (pr_bg + pr_fg)
        m = self.norm_batch(self.gauss(refine.reshape(-1, 1, H, W)).reshape(-1, H, W))
        num_fg = int(self.beta * H * W)
        top = torch.topk(m.reshape(B, -1), k=num_fg, dim=-1)[1]
        out = torch.zeros(B, H * W)
        out.scatter_(1, top, 1.0)
        return out.reshape(B, H, W), m

    def denorm(self, videos):
        std = torch.tensor(FRAME_STD).view(1, 3, 1, 1, 1)
        mean = torch.tensor(FRAME_MEAN).view(1, 3, 1, 1, 1)
        return videos * std + mean

    def getmask(self, clips):
        """fame.py:89-98."""
        B, C, T, H, W = clips.shape
        d = (clips[:, :, 0:-1] - clips[:, :, 1:]).abs().sum(dim=1).mean(dim=1)
        m = self.norm_batch(self.gauss(d.reshape(-1, 1, H, W)).reshape(-1, H, W))
        return self.get_seg(m, clips)

    def getmask_per_frame(self, clips):
        """fame.py:100-112: one mask per frame PAIR (i, i+1), i even; the colour model always uses the whole clip."""
        B, C, T, H, W = clips.shape
        out = []
        for i in range(0, T, 2):
            d = (clips[:, :, i] - clips[:, :, i + 1]).abs().sum(dim=1)
            m = self.norm_batch(self.gauss(d.reshape(-1, 1, H, W)).reshape(-1, H, W))
            out.append(self.get_seg(m, clips))
        return out

    def forward(self, videos, label, index, rand_batch):
        """fame.py:114-153 with the two random draws (`torch.randperm(B)`, `torch.rand(B)`) passed in.
        Returns (all_videos, all_label, (mask[B,196], masks_per_frame[B, T/2*196]), full-resolution binary masks, soft masks)."""
        B, C, T, H, W = videos.shape
        clips = self.denorm(videos.contiguous())
        mask, soft = self.getmask(clips)
        per = self.getmask_per_frame(clips)
        mpf = torch.stack([p[0] for p in per]).permute(1, 0, 2, 3)
        soft_pf = torch.stack([p[1] for p in per]).permute(1, 0, 2, 3)
        m5 = mask.to(videos.dtype).view(B, 1, 1, H, W)
        fuse = videos[index] * (1 - m5) + videos * m5
        if self.prob_aug < 1:
            aug = torch.where(rand_batch < self.prob_aug)[0]
            ori = torch.where(rand_batch >= self.prob_aug)[0]
            all_videos = torch.cat([fuse[aug], videos[ori]], 0)
            all_label = torch.cat([label[aug], label[ori]], 0)
            m5 = torch.cat([m5[aug], m5[ori]], 0)
            mpf = torch.cat([mpf[aug], mpf[ori]], 0)
        else:
            all_videos, all_label = fuse, label
        pooled = F.avg_pool2d(m5.squeeze(1).squeeze(1), kernel_size=16, stride=16).reshape(B, -1)
        pooled_pf = F.avg_pool2d(mpf, kernel_size=16, stride=16).reshape(B, -1)
        return all_videos, all_label, (pooled, pooled_pf), (mask, soft, soft_pf)
