"""Framework-independent synthetic weights / inputs.

Every value is a pure function of (seed, key string, flat index) through the
splitmix64 integer mix, evaluated with numpy uint64 arithmetic.  The same
formulae therefore give the same bits in the golden generator (which runs
beside the reference), in the CPU oracle tests and on the GPU box, so no
weight file ever has to be shipped.

Nothing here is on the hot path: it is host-side data generation.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step (vectorised over a uint64 array)."""
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def key_id(key: str) -> int:
    """FNV-1a 64-bit hash of a state_dict key (or input name)."""
    h = 0xCBF29CE484222325
    for b in key.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_u64(seed: int, key: str, n: int) -> np.ndarray:
    """h(seed, key_id, i) for i in [0, n) as uint64."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([(seed * 0x9E3779B97F4A7C15 + key_id(key)) & 0xFFFFFFFFFFFFFFFF],
                                   dtype=np.uint64))[0]
        i = np.arange(n, dtype=np.uint64)
        return splitmix64(base + i)


def uniform24(seed: int, key: str, n: int) -> np.ndarray:
    """u = (h >> 40) / 2^24 in [0, 1) as float64 (exact 24-bit dyadic)."""
    return (hash_u64(seed, key, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _t(a: np.ndarray, shape, dtype=torch.float32) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(shape).to(dtype)


def _is_layernorm(name: str) -> bool:
    # blocks.N.norm1/2, norm, agg_block...norm / norm_context, agg_block.last_layer.0
    parts = name.split(".")
    return any(p in ("norm", "norm1", "norm2", "norm_context", "fc_norm") for p in parts) or "last_layer" in name


def param_values(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Formula weights keyed on the parameter name."""
    n = int(np.prod(shape)) if len(shape) else 1
    s = 2.0 * uniform24(seed, name, n) - 1.0  # in [-1, 1), dyadic
    if name.endswith("latents"):
        v = math.sqrt(3.0) * s
    elif name in ("cls_token", "pos_embed"):
        v = 0.02 * s
    elif _is_layernorm(name):
        v = 1.0 + 0.1 * s if name.endswith("weight") else 0.01 * s
    elif name.endswith("weight") and len(shape) >= 2:
        v = np.round(0.02 * math.sqrt(3.0) * s * 65536.0) / 65536.0
    else:  # biases, q_bias, v_bias
        v = 0.01 * s
    return _t(v.astype(np.float32), shape)


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter (deduplicated names: tied layers share layer 0)."""
    for name, p in module.named_parameters():
        p.copy_(param_values(name, tuple(p.shape), seed).to(p.dtype))


def fill_params(shapes: dict, seed: int = 0) -> dict:
    return {k: param_values(k, tuple(s), seed) for k, s in shapes.items()}


def video(batch: int, frames: int, size: int, seed: int = 1000, first: int = 0) -> torch.Tensor:
    """x[B,3,T,H,W] = (floor(u*4096) - 2048)/1024 in [-2, 2).  Clip b is hashed on its
    global index first+b so that a DP shard equals the matching slice of the global batch."""
    per = 3 * frames * size * size
    out = np.empty((batch, per), dtype=np.float32)
    for b in range(batch):
        u = uniform24(seed, f"video.{first + b}", per)
        out[b] = ((np.floor(u * 4096.0) - 2048.0) / 1024.0).astype(np.float32)
    return _t(out, (batch, 3, frames, size, size))


def targets(batch: int, num_classes: int = 400, seed: int = 1000, first: int = 0) -> torch.Tensor:
    h = np.array([hash_u64(seed, f"target.{first + b}", 1)[0] for b in range(batch)], dtype=np.uint64)
    return torch.from_numpy((h % np.uint64(num_classes)).astype(np.int64))


def teacher_logits(batch: int, num_scene: int = 365, seed: int = 1000, first: int = 0) -> torch.Tensor:
    out = np.empty((batch, num_scene), dtype=np.float32)
    for b in range(batch):
        u = uniform24(seed, f"teacher.{first + b}", num_scene)
        out[b] = ((np.floor(u * 2048.0) - 1024.0) / 256.0).astype(np.float32)
    return _t(out, (batch, num_scene))


def fg_masks(batch: int, tokens: int, grid: int = 196, seed: int = 1000, first: int = 0):
    """(mask[B,grid], masks_per_frame[B,tokens]) with values k/256 (what FAME's 16x16
    average pooling of a binary mask produces, reference utils/transform/fame.py:142-148)."""
    m = np.empty((batch, grid), dtype=np.float32)
    mn = np.empty((batch, tokens), dtype=np.float32)
    for b in range(batch):
        m[b] = (np.floor(uniform24(seed, f"fg196.{first + b}", grid) * 257.0) / 256.0).astype(np.float32)
        mn[b] = (np.floor(uniform24(seed, f"fgN.{first + b}", tokens) * 257.0) / 256.0).astype(np.float32)
    return _t(m, (batch, grid)), _t(mn, (batch, tokens))


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def scene_video(batch: int, frames: int, size: int, seed: int = 2000, first: int = 0) -> torch.Tensor:
    """Structured clips for the FAME path, x[B,3,T,H,W] fp32, ImageNet-normalised: a static colour ramp background, a 60-pixel
    square moving 6 px/frame right and 4 px/frame down, 3 bits of hashed noise.  Integer arithmetic up to the final
    (v/256 - mean)/std in IEEE fp32, so every machine produces the same bits."""
    out = np.empty((batch, 3, frames, size, size), dtype=np.float32)
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    obj = (200, 40, 90)
    for b in range(batch):
        gb = first + b
        noise = (hash_u64(seed, f"scene.{gb}", 3 * frames * size * size) >> np.uint64(61)).astype(np.int64).reshape(3, frames, size, size)
        for c in range(3):
            base = (xx * (3 + c) + yy * (2 + (gb % 5) + c)) % 256
            for t in range(frames):
                x0, y0 = 20 + 10 * (gb % 7) + 6 * t, 30 + 4 * t
                inside = (xx >= x0) & (xx < x0 + 60) & (yy >= y0) & (yy < y0 + 60)
                v = np.where(inside, obj[(c + gb) % 3], base) + noise[c, t]
                v = np.minimum(v, 255).astype(np.float32) / np.float32(256.0)
                out[b, c, t] = (v - np.float32(IMAGENET_MEAN[c])) / np.float32(IMAGENET_STD[c])
    return torch.from_numpy(out)


def dropout_mask(kind: str, block: int, shape, keep: float, seed: int = 3000) -> torch.Tensor:
    """fp32 mask of 0 / (1 / keep) for an nn.Dropout / drop_path call of the encoder, as a formula (kept where u < keep): what the golden generator
    hands the reference's dropout modules and the tests hand the oracle and the HIP model.  kind: 'pos' | 'proj' | 'mlp' | 'path1' | 'path2'."""
    n = int(np.prod(shape))
    u = uniform24(seed, f"drop.{kind}.{block}", n)
    inv = np.float32(1.0) / np.float32(keep)
    return _t(np.where(u < keep, inv, np.float32(0)).astype(np.float32), tuple(shape))


def attn_drop_seed(block: int, seed: int = 3000) -> int:
    """64-bit seed of block `block`'s attention-matrix dropout mask (devias_mhsa_fwd_dropout)"""
    return int(hash_u64(seed, f"drop.attn.{block}", 1)[0])
