"""CPU ORACLE -- test infrastructure, NOT product code.

A plain-PyTorch (CPU, fp32/fp64) functional restatement of the DEVIAS slot-ViT
training step.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product path (devias_amd/) never
does and fails loudly when the HIP library is missing.

Pinned: tests/golden/make_goldens.py imports the real reference (in the build
container only), feeds it the formula weights/inputs of devias_amd/synth.py and
(a) asserts this restatement reproduces the reference's outputs, losses and
gradients, (b) commits those reference outputs as fixtures under tests/golden/.
tests/test_oracle_golden.py re-checks the oracle against those fixtures
everywhere (including the GPU box where /root/reference does not exist).

Each function cites the reference file:line it restates (paths relative to the
reference root).  The math fine print is SURVEY.md §9.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class SlotViTConfig:
    """Constructor arguments that matter (model/modeling_slot.py:222-250, :416-422)."""
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 400
    num_scene_classes: int = 365
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    all_frames: int = 16
    tubelet_size: int = 2
    num_latents: int = 2
    agg_depth: int = 8
    agg_weights_tie: bool = True
    agg_heads: int = 4          # agg_block/agg_block.py:83 (hard-coded)
    agg_dim_head: int = 512     # agg_block/agg_block.py:83 (hard-coded)
    agg_ff_mult: int = 4
    eps_encoder: float = 1e-6   # modeling_slot.py:420
    eps_agg: float = 1e-5       # nn.LayerNorm default, agg_block/attention.py:29-30
    mask_hidden: Tuple[int, int] = (512, 256)   # modeling_slot.py:199-203
    head_type: str = "linear"   # 'linear' (the recipes) or 'mlp' (MLPHead, hidden 512: modeling_slot.py:23-34, 307-313)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid * (self.all_frames // self.tubelet_size)

    @property
    def head_width(self) -> int:
        return self.num_classes + self.num_scene_classes


def param_shapes(cfg: SlotViTConfig) -> Dict[str, Tuple[int, ...]]:
    """The reference's named_parameters() (deduplicated; tied layers -> layer 0) -- SURVEY.md §8b."""
    D, Dff = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)
    inner = cfg.agg_heads * cfg.agg_dim_head
    s: Dict[str, Tuple[int, ...]] = {}
    s["patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,); s[p + "norm1.bias"] = (D,)
        s[p + "attn.q_bias"] = (D,); s[p + "attn.v_bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,); s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Dff, D); s[p + "mlp.fc1.bias"] = (Dff,)
        s[p + "mlp.fc2.weight"] = (D, Dff); s[p + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    s["agg_block.latents"] = (cfg.num_latents, D)
    for l in range(1 if cfg.agg_weights_tie else cfg.agg_depth):
        p = f"agg_block.layers.{l}."
        s[p + "0.fn.to_q.weight"] = (inner, D)
        s[p + "0.fn.to_k.weight"] = (inner, D)
        s[p + "0.fn.to_v.weight"] = (inner, D)
        s[p + "0.fn.to_out.0.weight"] = (D, inner); s[p + "0.fn.to_out.0.bias"] = (D,)
        s[p + "0.norm.weight"] = (D,); s[p + "0.norm.bias"] = (D,)
        s[p + "0.norm_context.weight"] = (D,); s[p + "0.norm_context.bias"] = (D,)
        s[p + "2.fn.net.0.weight"] = (cfg.agg_ff_mult * D, D); s[p + "2.fn.net.0.bias"] = (cfg.agg_ff_mult * D,)
        s[p + "2.fn.net.3.weight"] = (D, cfg.agg_ff_mult * D); s[p + "2.fn.net.3.bias"] = (D,)
        s[p + "2.norm.weight"] = (D,); s[p + "2.norm.bias"] = (D,)
    s["agg_block.last_layer.0.weight"] = (D,); s["agg_block.last_layer.0.bias"] = (D,)
    h1, h2 = cfg.mask_hidden
    s["mask_predictor.decoder.0.weight"] = (h1, D); s["mask_predictor.decoder.0.bias"] = (h1,)
    s["mask_predictor.decoder.2.weight"] = (h2, h1); s["mask_predictor.decoder.2.bias"] = (h2,)
    s["mask_predictor.decoder.4.weight"] = (cfg.grid * cfg.grid, h2)
    s["mask_predictor.decoder.4.bias"] = (cfg.grid * cfg.grid,)
    if cfg.head_type == "mlp":
        s["head.fc1.weight"] = (512, D); s["head.fc1.bias"] = (512,)
        s["head.fc2.weight"] = (cfg.head_width, 512); s["head.fc2.bias"] = (cfg.head_width,)
    else:
        s["head.weight"] = (cfg.head_width, D); s["head.bias"] = (cfg.head_width,)
    return s


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """model/modeling_slot.py:181-191 -- float64 numpy table, cast to fp32, shape [1,N,D]."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    table = angle.copy()
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.tensor(table, dtype=torch.float32).unsqueeze(0)


def patch_embed(P, cfg: SlotViTConfig, x: torch.Tensor) -> torch.Tensor:
    """PatchEmbed.forward model/modeling_slot.py:171-177 restated as the GEMM of SURVEY.md §9:
    token n=(t'*g+h')*g+w', feature f=((c*2+kt)*16+kh)*16+kw."""
    B, C, T, H, W = x.shape
    assert H == cfg.img_size and W == cfg.img_size, "Input image size doesn't match model"
    ts, ps, g = cfg.tubelet_size, cfg.patch_size, cfg.grid
    a = x.reshape(B, C, T // ts, ts, g, ps, g, ps).permute(0, 2, 4, 6, 1, 3, 5, 7)
    a = a.reshape(B, (T // ts) * g * g, C * ts * ps * ps)
    w = P["patch_embed.proj.weight"].reshape(cfg.embed_dim, -1)
    return a @ w.t() + P["patch_embed.proj.bias"]


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    M = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    return x


def attn_drop_mask(keep: float, seed: int, B: int, H: int, N: int) -> torch.Tensor:
    """The mask devias_mhsa_fwd_dropout / _bwd_dropout apply to the softmax matrix (include/devias_amd.h, csrc/attention.hip drop_rowkey / drop_scale),
    restated in numpy: [B, H, N, N] fp32 of 0 or 1 / keep.  It stands in for nn.Dropout(attn_drop)'s mask (model/modeling_slot.py:90,110), which the
    reference draws from torch's generator: tests/golden/make_goldens.py hands THIS mask to the reference's F.dropout call."""
    M = np.uint64(0xFFFFFFFF)
    keep32 = np.float32(keep)
    t = np.floor(np.float64(keep32) * 4294967296.0)
    thresh = np.uint64(4294967295 if t >= 4294967295.0 else int(t))
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    bh = np.arange(B * H, dtype=np.uint64)
    a = _mix32(s0 ^ ((bh * np.uint64(0x9E3779B1)) & M))                                        # [BH]
    i = np.arange(N, dtype=np.uint64)
    rowkey = _mix32((a[:, None] + s1 + ((i * np.uint64(0x85EBCA6B)) & M)[None, :]) & M)       # [BH, N]
    u = _mix32((rowkey[:, :, None] + ((i * np.uint64(0xC2B2AE35)) & M)[None, None, :]) & M)    # [BH, N(query), N(key)]
    inv = np.float32(1.0) / keep32
    return torch.from_numpy(np.where(u < thresh, inv, np.float32(0)).astype(np.float32).reshape(B, H, N, N))


def encoder_block(P, cfg: SlotViTConfig, i: int, x: torch.Tensor, drop: Optional[dict] = None) -> torch.Tensor:
    """Block.forward model/modeling_slot.py:142-152 with Attention.forward :92-117 and Mlp :60-67.
    `drop` (training with drop_rate / attn_drop_rate / drop_path > 0) makes the reference's random masks explicit, each already scaled by 1 / keep:
    'attn' [B, H, N, N] (attn_drop :110), 'proj' [B, N, D] (proj_drop :114), 'mlp' [B, N, D] (Mlp.drop :66), 'path1' / 'path2' [B] (drop_path :150-151)."""
    drop = drop or {}
    bcast = lambda m: m.to(x.dtype).reshape(-1, 1, 1)   # noqa: E731
    p = f"blocks.{i}."
    B, N, D = x.shape
    H = cfg.num_heads
    dh = D // H
    u = F.layer_norm(x, (D,), P[p + "norm1.weight"], P[p + "norm1.bias"], cfg.eps_encoder)
    bias = torch.cat([P[p + "attn.q_bias"], torch.zeros_like(P[p + "attn.v_bias"]), P[p + "attn.v_bias"]])
    qkv = F.linear(u, P[p + "attn.qkv.weight"], bias).reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * dh ** -0.5, qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    if drop.get("attn") is not None:
        attn = attn * drop["attn"].to(attn.dtype)
    o = (attn @ v).transpose(1, 2).reshape(B, N, D)
    y = F.linear(o, P[p + "attn.proj.weight"], P[p + "attn.proj.bias"])
    if drop.get("proj") is not None:
        y = y * drop["proj"].to(y.dtype)
    if drop.get("path1") is not None:
        y = y * bcast(drop["path1"])
    x = x + y
    u2 = F.layer_norm(x, (D,), P[p + "norm2.weight"], P[p + "norm2.bias"], cfg.eps_encoder)
    h = F.gelu(F.linear(u2, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]))
    y = F.linear(h, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
    if drop.get("mlp") is not None:
        y = y * drop["mlp"].to(y.dtype)
    if drop.get("path2") is not None:
        y = y * bcast(drop["path2"])
    return x + y


def forward_features(P, cfg: SlotViTConfig, x: torch.Tensor, taps: Optional[dict] = None, drops: Optional[dict] = None) -> torch.Tensor:
    """VisionTransformer.forward_features model/modeling_slot.py:350-377.  `drops`: {'pos': [B, N, D] mask of pos_drop (:356), i: encoder_block's `drop` of block i}."""
    drops = drops or {}
    x = patch_embed(P, cfg, x)
    x = x + sinusoid_table(cfg.num_patches, cfg.embed_dim).to(x.dtype)
    if drops.get("pos") is not None:
        x = x * drops["pos"].to(x.dtype)
    for i in range(cfg.depth):
        x = encoder_block(P, cfg, i, x, drops.get(i))
        if taps is not None:
            taps[f"block{i}"] = x
    D = cfg.embed_dim
    return F.layer_norm(x, (D,), P["norm.weight"], P["norm.bias"], cfg.eps_encoder)


def agg_block(P, cfg: SlotViTConfig, feats: torch.Tensor, taps: Optional[dict] = None):
    """AggregationBlock.forward agg_block/agg_block.py:120-139; PreNorm agg_block/attention.py:32-40;
    slot Attention.forward agg_block/attention.py:120-141 (softmax over the SLOT axis :132, token
    renormalisation :136); FeedForward :81-82."""
    B, N, D = feats.shape
    S, h, dh = cfg.num_latents, cfg.agg_heads, cfg.agg_dim_head
    xs = P["agg_block.latents"].unsqueeze(0).expand(B, S, D).to(feats.dtype)
    A = None
    for l in range(cfg.agg_depth):
        p = f"agg_block.layers.{0 if cfg.agg_weights_tie else l}."
        qn = F.layer_norm(xs, (D,), P[p + "0.norm.weight"], P[p + "0.norm.bias"], cfg.eps_agg)
        c = F.layer_norm(feats, (D,), P[p + "0.norm_context.weight"], P[p + "0.norm_context.bias"], cfg.eps_agg)
        q = F.linear(qn, P[p + "0.fn.to_q.weight"]).reshape(B, S, h, dh).permute(0, 2, 1, 3)   # [B,h,S,dh]
        k = F.linear(c, P[p + "0.fn.to_k.weight"]).reshape(B, N, h, dh).permute(0, 2, 1, 3)    # [B,h,N,dh]
        v = F.linear(c, P[p + "0.fn.to_v.weight"]).reshape(B, N, h, dh).permute(0, 2, 1, 3)
        sim = (q @ k.transpose(-1, -2)) * dh ** -0.5        # [B,h,S,N]
        A = sim.softmax(dim=2)                              # over the slot axis
        An = A / (A.sum(dim=-1, keepdim=True) + 1e-7)
        o = (An @ v).permute(0, 2, 1, 3).reshape(B, S, h * dh)
        xs = F.linear(o, P[p + "0.fn.to_out.0.weight"], P[p + "0.fn.to_out.0.bias"]) + xs
        f = F.layer_norm(xs, (D,), P[p + "2.norm.weight"], P[p + "2.norm.bias"], cfg.eps_agg)
        f = F.gelu(F.linear(f, P[p + "2.fn.net.0.weight"], P[p + "2.fn.net.0.bias"]))
        xs = F.linear(f, P[p + "2.fn.net.3.weight"], P[p + "2.fn.net.3.bias"]) + xs
        if taps is not None:
            taps[f"agg{l}"] = xs
    slots = F.layer_norm(xs, (D,), P["agg_block.last_layer.0.weight"], P["agg_block.last_layer.0.bias"], cfg.eps_agg)
    return slots, A.reshape(B * h, S, N)


def mask_predictor(P, cfg: SlotViTConfig, slots_flat: torch.Tensor) -> torch.Tensor:
    """MaskPredictor.forward model/modeling_slot.py:209-216."""
    m = F.relu(F.linear(slots_flat, P["mask_predictor.decoder.0.weight"], P["mask_predictor.decoder.0.bias"]))
    m = F.relu(F.linear(m, P["mask_predictor.decoder.2.weight"], P["mask_predictor.decoder.2.bias"]))
    return torch.sigmoid(F.linear(m, P["mask_predictor.decoder.4.weight"], P["mask_predictor.decoder.4.bias"]))


def student_forward(P, cfg: SlotViTConfig, x: torch.Tensor, taps: Optional[dict] = None, drops: Optional[dict] = None):
    """VisionTransformer.forward ('matching' branch) model/modeling_slot.py:379-410.  `drops`: forward_features' masks (training with dropout)."""
    feats = forward_features(P, cfg, x, taps, drops)
    slots, attn = agg_block(P, cfg, feats, taps)
    B, S, D = slots.shape
    slots_flat = slots.reshape(-1, D)
    if cfg.head_type == "mlp":                                            # MLPHead: fc2(relu(fc1(x))), modeling_slot.py:30-33
        slots_head = F.linear(F.relu(F.linear(slots_flat, P["head.fc1.weight"], P["head.fc1.bias"])), P["head.fc2.weight"], P["head.fc2.bias"])
    else:
        slots_head = F.linear(slots_flat, P["head.weight"], P["head.bias"])
    probs = F.softmax(slots_head, dim=-1).view(B, S, -1)
    nb = cfg.num_classes
    a_idx = torch.argmax(probs[:, :, :nb].max(dim=-1).values, dim=1)
    s_idx = torch.argmax(probs[:, :, nb:nb + cfg.num_scene_classes].max(dim=-1).values, dim=1)
    ar = torch.arange(B)
    action_feat, scene_feat = slots[ar, a_idx], slots[ar, s_idx]
    action_logit = slots_head.view(B, S, -1)[ar, a_idx]
    scene_logit = slots_head.view(B, S, -1)[ar, s_idx]
    mask_predictions = mask_predictor(P, cfg, slots_flat)
    return (action_feat, scene_feat), (action_logit, scene_logit, attn), (slots_head, slots_flat, mask_predictions)


def match_slots(cost: torch.Tensor):
    """argmin over ordered pairs i != j of cost[i,0] + cost[j,1]; equals
    scipy.optimize.linear_sum_assignment on the S x 2 matrix (utils/loss/train_loss.py:112-122).
    Ties resolve to the lexicographically first (i, j), which is what SciPy returns for S=2."""
    S = cost.shape[0]
    best, bi, bj = None, 0, 1
    for i, j in itertools.permutations(range(S), 2):
        c = float(cost[i, 0]) + float(cost[j, 1])
        if best is None or c < best:
            best, bi, bj = c, i, j
    return bi, bj


def train_loss(cfg: SlotViTConfig, student_output, teacher_scene_logit: torch.Tensor, target: torch.Tensor,
               fg_mask, scene_loss_weight: float = 4000.0, mask_prediction_loss_weight: float = 1.0,
               mask_distill_loss_weight: float = 1.0, scene_criterion: str = "KL"):
    """TrainLoss.forward 'matching' branch, scene_criterion 'KL' (the recipe) or 'CE' -- utils/loss/train_loss.py:85-187.
    Returns (total_loss[1], matched action logits [B,C], dict of 5 floats, (i*, j*) index tensors)."""
    _, (_, _, attn), (slots_head, slots, mask_predictions) = student_output
    bs = target.shape[0]
    S = slots_head.shape[0] // bs
    nh = attn.shape[0] // bs
    C = slots_head.shape[1]
    nb = cfg.num_classes
    Ahat = attn.reshape(bs, nh, S, -1).mean(dim=1)                       # :97
    M = mask_predictions.reshape(bs, S, -1)
    scene_target = torch.argmax(teacher_scene_logit, dim=1) + nb         # :100, :107
    pad = teacher_scene_logit.min() - 1.0                                # :103
    Tpad = torch.cat([torch.full((bs, nb), float(pad), dtype=teacher_scene_logit.dtype), teacher_scene_logit], dim=1)
    p = slots_head.softmax(-1).detach().reshape(bs, S, C)                # :109
    Z = slots_head.view(bs, S, C)
    fg196, fgN = fg_mask
    act = slots_head.new_zeros(1); scn = slots_head.new_zeros(1)
    mp = slots_head.new_zeros(1); md = slots_head.new_zeros(1)
    rows, ii, jj = [], [], []
    for b in range(bs):
        cost = torch.stack([-p[b, :, target[b]], -p[b, :, scene_target[b]]], dim=1)   # :112-118
        i, j = match_slots(cost)
        ii.append(i); jj.append(j)
        md = md + F.mse_loss(Ahat[b, i], fgN[b]) * mask_distill_loss_weight                       # :145
        mp = mp + F.binary_cross_entropy_with_logits(M[b, i], fg196[b]) * mask_prediction_loss_weight  # :146-149
        act = act + F.cross_entropy(Z[b, i], target[b])                                           # :150
        rows.append(Z[b, i])
        if scene_criterion == "CE":
            scn = scn + F.cross_entropy(Z[b, j], scene_target[b])                                  # :155-156 (no scene_loss_weight)
        else:
            scn = scn + F.kl_div(F.log_softmax(Z[b, j], dim=-1), F.log_softmax(Tpad[b], dim=-1),
                                 reduction="batchmean", log_target=True) * scene_loss_weight        # :159-164
    act, scn, mp, md = act / bs, scn / bs, mp / bs, md / bs                                        # :168-171
    sl = F.normalize(slots.reshape(bs, S, -1), p=2, dim=2)                                         # :173-178
    cs = torch.bmm(sl, sl.transpose(1, 2)) * (1 - torch.eye(S, dtype=sl.dtype))
    cos = (cs.sum(dim=(1, 2)) / (S * (S - 1))).mean()
    total = act + scn + cos + mp + md                                                              # :180
    ld = {"action_loss": act.item(), "scene_loss": scn.item(), "cosine_loss": cos.item(),
          "mask_prediction_loss": mp.item(), "mask_distill_loss": md.item()}
    return total, torch.stack(rows), ld, (torch.tensor(ii), torch.tensor(jj))


# --------------------------------------------------------------------------------------------
# teacher (model/modeling_finetune.py:178-325, use_mean_pooling=False) -- "next" row §8f-1
# --------------------------------------------------------------------------------------------
def teacher_param_shapes(cfg: SlotViTConfig, num_classes: int = 365) -> Dict[str, Tuple[int, ...]]:
    D, Dff = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)
    s: Dict[str, Tuple[int, ...]] = {"cls_token": (1, 1, D)}
    s["patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,); s[p + "norm1.bias"] = (D,)
        s[p + "attn.q_bias"] = (D,); s[p + "attn.v_bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,); s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Dff, D); s[p + "mlp.fc1.bias"] = (Dff,)
        s[p + "mlp.fc2.weight"] = (D, Dff); s[p + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    s["head.weight"] = (num_classes, D); s["head.bias"] = (num_classes,)
    return s


def teacher_forward(P, cfg: SlotViTConfig, x: torch.Tensor):
    """modeling_finetune.VisionTransformer.forward with use_mean_pooling=False
    (model/modeling_finetune.py:273-325): patch-embed -> cat(cls_token, x) (cls FIRST, :277-279) ->
    + sinusoid pos over N+1 positions (:282-283) -> blocks -> norm -> token = x[:, 0] (:308) -> head.
    Returns (token, logits)."""
    x = patch_embed(P, cfg, x)
    B = x.shape[0]
    x = torch.cat([P["cls_token"].expand(B, -1, -1).to(x.dtype), x], dim=1)
    x = x + sinusoid_table(cfg.num_patches + 1, cfg.embed_dim).to(x.dtype)
    for i in range(cfg.depth):
        x = encoder_block(P, cfg, i, x)
    x = F.layer_norm(x, (cfg.embed_dim,), P["norm.weight"], P["norm.bias"], cfg.eps_encoder)
    tok = x[:, 0]
    return tok, F.linear(tok, P["head.weight"], P["head.bias"])


def train_step(P, cfg: SlotViTConfig, x, target, teacher_scene_logit, fg_mask, drops: Optional[dict] = None, **loss_kw):
    """train_class_batch engine/engine_for_slot.py:50-56 (teacher logits supplied) + backward.
    Returns (total, logits, loss_dict, grads: name -> tensor, student_output).  `drops`: forward_features' dropout masks."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = student_forward(Pg, cfg, x, None, drops)
    total, logits, ld, idx = train_loss(cfg, out, teacher_scene_logit, target, fg_mask, **loss_kw)
    total.backward()
    grads = {k: v.grad for k, v in Pg.items()}
    return total.detach(), logits.detach(), ld, grads, out, idx
