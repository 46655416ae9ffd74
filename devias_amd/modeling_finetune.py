"""Frozen scene teacher (model/modeling_finetune.py:178-334 of the reference, `vit_base_patch16_224` with
use_mean_pooling=False): cls token + N patch tokens, forward only, on the same HIP kernels as the student.

`forward(x, return_attn=False)` returns `(token[B, D], logits[B, num_classes])` like the reference (:314-325); it is what
`engine/engine_for_slot.train_class_batch` calls under `torch.no_grad()` (:52-53).  N + 1 = 1569 tokens per clip is odd, so the
token matrix is padded (once, with zero rows at the END of the batch) to a multiple of 256 rows to stay on the full-tile GEMM
kernels; the attention kernels handle the ragged sequence length themselves."""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from . import ops
from .modeling_slot import (ACT_GELU, Block, PatchEmbed, _WCACHE, _cfg, _f32, _qkv_bias, get_sinusoid_encoding_table, register_model)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, fc_drop_rate=0., drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, init_scale=0.,
                 all_frames=16, tubelet_size=2, use_checkpoint=False, use_mean_pooling=True, compute_dtype='bf16'):
        super().__init__()
        if use_mean_pooling:
            raise NotImplementedError("only the cls-token teacher (use_mean_pooling=False) is on the DEVIAS slot path "
                                      "(run_slot_finetuning.py:392-406)")
        if use_learnable_pos_emb or fc_drop_rate or drop_rate or attn_drop_rate:
            raise NotImplementedError("learnable pos-emb / dropout are not used by the frozen teacher")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.tubelet_size = tubelet_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      num_frames=all_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.pos_embed = get_sinusoid_encoding_table(num_patches, embed_dim)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  norm_layer=norm_layer, init_values=init_values) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc_norm = None
        self.fc_dropout = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.head.weight, std=.02)
        self.apply(self._init_weights)
        self.head.weight.data.mul_(init_scale)
        self.head.bias.data.mul_(init_scale)
        self.compute_dtype = {'bf16': torch.bfloat16, 'fp32': torch.float32}[compute_dtype]
        self._pos_cache = {}

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def set_compute_dtype(self, compute_dtype):
        self.compute_dtype = {'bf16': torch.bfloat16, 'fp32': torch.float32}[compute_dtype]
        return self

    def _pos(self, device, dtype):
        key = (str(device), dtype)
        if key not in self._pos_cache:
            self._pos_cache[key] = self.pos_embed[0].to(device=device, dtype=dtype).contiguous()
        return self._pos_cache[key]

    @torch.no_grad()
    def forward_features(self, x, return_attn=False):
        if return_attn:
            raise NotImplementedError("per-block attention maps are never materialised by the fused MHSA kernel")
        if not x.is_cuda:
            raise RuntimeError("devias_amd teacher runs on an MI355X only (HIP kernels; no CPU fallback)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        B = x.shape[0]
        pe = self.patch_embed
        N, D, cdt = pe.num_patches, self.embed_dim, self.compute_dtype
        Nt = N + 1
        pos = self._pos(x.device, cdt)                                   # [N+1, D]
        A = ops.patch_im2col(x, pe.tubelet_size, pe.patch_size[0], cdt)
        tok = ops.gemm(A, _WCACHE.get(pe.proj.weight, cdt), bias=_f32(pe.proj.bias), res=pos[1:].contiguous(), res_mod=N)
        cls = (self.cls_token.detach().reshape(1, D).float() + pos[:1].float()).to(cdt)        # cls FIRST, then + pos (:277-283)
        M = B * Nt
        Mp = ((M + 255) // 256) * 256
        h = torch.zeros((Mp, D), dtype=cdt, device=x.device)
        hv = h[:M].view(B, Nt, D)
        hv[:, 0] = cls
        hv[:, 1:] = tok.view(B, N, D)
        scale = 64 ** -0.5
        # attention output buffer, padded like h: allocated ONCE, mhsa_fwd writes its first M rows in place every block and the pad rows
        # stay zero (no per-block zeros() + slice copy)
        o = torch.zeros((Mp, D), dtype=cdt, device=x.device) if Mp != M else torch.empty((M, D), dtype=cdt, device=x.device)
        for blk in self.blocks:
            a = blk.attn
            Wqkv, Wp, W1, W2 = (_WCACHE.get(w, cdt) for w in (a.qkv.weight, a.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight))
            u, _, _ = ops.layernorm_fwd(h, _f32(blk.norm1.weight), _f32(blk.norm1.bias), blk.norm1.eps)
            qkv = ops.gemm(u, Wqkv, bias=_qkv_bias(a.q_bias, a.v_bias))          # q_bias | 0 | v_bias, rebuilt only when a bias changed
            ops.mhsa_fwd(qkv[:M], B, Nt, a.num_heads, scale, out=o[:M])
            h1 = ops.gemm(o, Wp, bias=_f32(a.proj.bias), res=h)
            u2, _, _ = ops.layernorm_fwd(h1, _f32(blk.norm2.weight), _f32(blk.norm2.bias), blk.norm2.eps)
            act = ops.gemm(u2, W1, bias=_f32(blk.mlp.fc1.bias), act=ACT_GELU)
            h = ops.gemm(act, W2, bias=_f32(blk.mlp.fc2.bias), res=h1)
        tokens0 = h[:M].view(B, Nt, D)[:, 0].contiguous()                # only the cls rows need the final LayerNorm
        y, _, _ = ops.layernorm_fwd(tokens0, _f32(self.norm.weight), _f32(self.norm.bias), self.norm.eps)
        return y

    @torch.no_grad()
    def forward(self, x, return_attn=False):
        token = self.forward_features(x, return_attn)
        logits = ops.gemm(token, _WCACHE.get(self.head.weight, self.compute_dtype), bias=_f32(self.head.bias))
        return token, logits


@register_model
def vit_base_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model
