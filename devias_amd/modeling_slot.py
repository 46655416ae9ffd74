"""MI355X-native slot-DEVIAS student model behind the reference's timm-style surface.

Mirrors model/modeling_slot.py + agg_block/{agg_block,attention}.py of the reference: same constructor kwargs,
same module tree / parameter names (state_dict compatible), same `forward()` 3-tuple.  The nn.Module
objects below (nn.Linear, nn.LayerNorm, nn.Conv3d ...) are PARAMETER CONTAINERS only: their own forward() is never
called.  All arithmetic runs in libdevias_amd.so (hand-written gfx950 HIP kernels) through four autograd Functions,
one per fused region:

    PatchEmbedFn   tubelet im2col -> GEMM(+bias +sinusoid pos)                       (modeling_slot.py:171-177, :354-355)
    EncoderBlockFn LN -> QKV GEMM -> flash MHSA -> proj GEMM(+res) -> LN -> fc1 GEMM(+GELU) -> fc2 GEMM(+res)   (:142-152)
    AggBlockFn     final LN -> [context LN -> K|V GEMM once per distinct weight set] -> depth x slot layer -> LN
                   (agg_block/agg_block.py:120-139, agg_block/attention.py:32-40,120-141)
    HeadFn         shared head GEMM + MaskPredictor MLP (ReLU/ReLU/Sigmoid epilogues)     (modeling_slot.py:392-393, :209-216)

`compute_dtype` selects the activation/weight storage type of the kernels: 'fp32' (parity mode: exact fp32 MFMA / VALU
kernels) or 'bf16' (measured mode: bf16 storage, fp32 accumulation and statistics).  Master parameters are always fp32.
There is no PyTorch fallback: without a GPU + the built HIP library forward() raises.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops
from .ops import ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID

_MODEL_REGISTRY: Dict[str, callable] = {}


def register_model(fn):
    """timm.models.registry.register_model stand-in (timm is used when importable, see create_model)."""
    _MODEL_REGISTRY[fn.__name__] = fn
    try:  # register with timm too so `timm.create_model('slot_vit_base_patch16_224', ...)` works as in the reference
        from timm.models.registry import register_model as _timm_register  # type: ignore
        _timm_register(fn)
    except Exception:
        pass
    return fn


def create_model(name: str, pretrained: bool = False, **kwargs):
    """timm.create_model look-alike: drops None kwargs like timm does (run_slot_finetuning.py:371-390)."""
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _MODEL_REGISTRY[name](pretrained=pretrained, **kwargs)


def _cfg(url="", **kwargs):
    return {"url": url, "num_classes": 400, "input_size": (3, 224, 224), "pool_size": None, "crop_pct": .9,
            "interpolation": "bicubic", "mean": (0.5, 0.5, 0.5), "std": (0.5, 0.5, 0.5), **kwargs}


def get_sinusoid_encoding_table(n_position: int, d_hid: int) -> torch.Tensor:
    """float64 table cast to fp32, [1, N, D] (modeling_slot.py:181-191)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    table = angle.copy()
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.tensor(table, dtype=torch.float32).unsqueeze(0)


# =====================================================================================================
# compute-dtype weight copies (bf16 mode): one cast per parameter per WEIGHT UPDATE
# =====================================================================================================
# A copy is valid while (weights epoch, Tensor._version, storage address, device) are unchanged.  `_version` catches every in-place
# update made through autograd-visible tensors (torch optimizers, load_state_dict, EMA copy_); updates made behind autograd's back --
# the fused optimizer writes parameters through raw pointers in the C ABI, and `.data` writes such as broadcast_parameters -- do not
# bump it, so those call invalidate_weight_cache(), which advances the global epoch.  Entries are keyed on the Parameter OBJECT
# through a WeakKeyDictionary: they die with their model, and an id()/address reused by a later model cannot hit them.
import weakref

_WEIGHTS_EPOCH = [0]


def invalidate_weight_cache() -> None:
    """Call after changing parameter VALUES in a way `Tensor._version` cannot see (raw-pointer kernels, `.data` writes)."""
    _WEIGHTS_EPOCH[0] += 1


class _WeakIdDict:
    """object -> value, keyed on IDENTITY and holding the key weakly (WeakKeyDictionary compares keys with ==, which is elementwise for
    tensors).  An entry disappears when its key object dies, so a recycled id() can never alias it."""

    def __init__(self):
        self._d = {}

    def get(self, obj, default=None):
        e = self._d.get(id(obj))
        return e[1] if e is not None and e[0]() is obj else default

    def __setitem__(self, obj, value):
        k = id(obj)
        self._d[k] = (weakref.ref(obj, lambda _r, k=k, d=self._d: d.pop(k, None) if (d.get(k) is not None and d[k][0] is _r) else None), value)

    def __len__(self):
        return len(self._d)


class _WeightCache:
    def __init__(self):
        self._c = _WeakIdDict()                      # Parameter -> (stamp, compute-dtype copy)
        self._cat = _WeakIdDict()                    # first Parameter of a concatenation -> {ids of the others: (stamps, weakrefs, copy)}
        self._qkvb = _WeakIdDict()                   # q_bias Parameter -> (stamps, weakref to v_bias, fp32 q_bias | 0 | v_bias)
        self._t = _WeakIdDict()                      # Parameter -> (stamp, transposed compute-dtype copy)
        self._qs = _WeakIdDict()                     # qkv weight Parameter / fp32 qkv bias tensor -> (stamp, copy with the q third scaled)
        self.casts = 0                               # number of casts performed (tests)
        self.transposes = 0
        self.qscaled = 0                             # q-scaled copies made (tests)

    @staticmethod
    def _stamp(p: torch.Tensor):
        return (_WEIGHTS_EPOCH[0], p._version, p.data_ptr(), p.device)

    def get(self, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        d = p.detach()
        if d.dim() > 2:
            d = d.reshape(d.shape[0], -1)
        if dtype == torch.float32:
            return d if d.is_contiguous() else d.contiguous()
        stamp = self._stamp(p)
        hit = self._c.get(p)
        if hit is not None and hit[0] == stamp and hit[1].dtype == dtype:
            return hit[1]
        w = ops.cast(d.contiguous(), dtype)
        self.casts += 1
        self._c[p] = (stamp, w)
        return w

    def get_t(self, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """the TRANSPOSE of a 2-D weight in the compute dtype ([in, out] for an nn.Linear weight [out, in]), cached like get(): what the encoder block's dgrad GEMMs
        read (devias_block_args.W*T) -- dY W with W transposed is a product of two k-contiguous operands, the fast layout of the GEMM kernels (round 6: 9-17 % less
        K-loop time than reading W through transposing LDS loads; same bits).  One extra copy per weight and weight update; bf16 mode only (callers)."""
        stamp = self._stamp(p)
        hit = self._t.get(p)
        if hit is not None and hit[0] == stamp and hit[1].dtype == dtype:
            return hit[1]
        w = self.get(p, dtype).t().contiguous()
        self.transposes += 1
        self._t[p] = (stamp, w)
        return w

    def get_qscaled(self, p: torch.Tensor, rows: int, factor: float, dtype: torch.dtype) -> torch.Tensor:
        """the compute-dtype copy of a [3D, D] qkv weight (or a [3D] fp32 qkv bias: dtype = torch.float32) whose first `rows` rows / entries -- the q third -- are multiplied
        by `factor` = scale * log2(e) IN FP32, before the rounding: the qkv GEMM then leaves q' = q * scale * log2(e) rounded ONCE, and forward, dQ and dK / dV kernels
        multiply the same bf16 operands (DEVIAS_ATTN_Q_PRESCALED, devias_block_args.WqkvS).  Cached like get(); backward reads the unscaled copies."""
        stamp = self._stamp(p) + (rows, factor)
        hit = self._qs.get(p)
        if hit is not None and hit[0] == stamp and hit[1].dtype == dtype:
            return hit[1]
        d = p.detach().float()
        d = d.reshape(d.shape[0], -1).clone() if d.dim() > 1 else d.clone()
        d[:rows] *= factor
        w = d if dtype == torch.float32 else ops.cast(d.contiguous(), dtype)
        self.qscaled += 1
        self._qs[p] = (stamp, w)
        return w

    def get_cat(self, ps, dtype: torch.dtype) -> torch.Tensor:
        """row-concatenation of several weights (to_k | to_v) in the compute dtype, cached like get()."""
        stamps = tuple(self._stamp(p) for p in ps)
        slot = self._cat.get(ps[0])
        key = tuple(id(p) for p in ps[1:])
        if slot is not None:
            hit = slot.get(key)
            if hit is not None and hit[0] == stamps and hit[2].dtype == dtype and all(r() is q for r, q in zip(hit[1], ps[1:])):
                return hit[2]
        w = torch.cat([self.get(p, dtype) for p in ps], dim=0).contiguous()
        if slot is None:
            slot = self._cat[ps[0]] = {}
        slot[key] = (stamps, tuple(weakref.ref(q) for q in ps[1:]), w)
        return w


_WCACHE = _WeightCache()


# column sums of a residual-stream gradient, produced for free by the LayerNorm-backward kernel that wrote it and consumed by the
# next backward region as the bias gradient of its last Linear.  The sum travels ON the gradient tensor object (autograd hands the
# same object to the next node when there is a single consumer) together with the tensor's storage address and VERSION COUNTER at the
# time of publication.  Two ways the sum could go stale are both caught: autograd builds a new tensor (hooks, out-of-place
# accumulation): the attribute is absent; autograd accumulates a second consumer's gradient IN PLACE into the tagged tensor
# (InputBuffer does that when the residual stream feeds two consumers): `_version` has moved.  In both cases the sum is recomputed.
_COLSUM_STATS = {"hit": 0, "miss": 0}


def _region_state(ctx, who: str, x: torch.Tensor):
    """The fused-region Functions keep their arena, weight copies and argument struct on `ctx` as raw pointers (not save_for_backward: the arena is
    written by the library, not by autograd) and drop them after ONE backward.  A second backward (retain_graph=True) or an input modified in place
    between forward and backward must fail with a message, not with a TypeError on None or silently wrong gradients."""
    if ctx.keep is None:
        raise RuntimeError(f"{who}: backward already consumed this region's saved arena; the fused regions support ONE backward per forward "
                           "(run forward again instead of retain_graph=True)")
    if x._version != ctx.x_version:
        raise RuntimeError(f"{who}: the region's input was modified in place between forward and backward (version {ctx.x_version} -> {x._version}); "
                           "its saved activations no longer match it")


def _publish_colsum(dx: torch.Tensor, cs: torch.Tensor) -> None:
    dx._devias_colsum = (cs, dx.data_ptr(), dx._version)


def _peek_colsum(dy: torch.Tensor) -> Optional[torch.Tensor]:
    """the published column sums of `dy` if they are still those of its current contents, else None (the tag is consumed either way)"""
    tag = getattr(dy, "_devias_colsum", None)
    if tag is None:
        _COLSUM_STATS["miss"] += 1
        return None
    del dy._devias_colsum
    cs, ptr, ver = tag
    if ptr == dy.data_ptr() and ver == dy._version and cs.numel() == dy.shape[1] and cs.device == dy.device:
        _COLSUM_STATS["hit"] += 1
        return cs
    _COLSUM_STATS["miss"] += 1
    return None


def _take_colsum(dy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    cs = _peek_colsum(dy)
    if cs is not None:
        if out is not None:
            out.copy_(cs)
            return out
        return cs
    return ops.colsum(dy, out=out)


# ---- gradient destinations ---------------------------------------------------------------------------------------------------------
# A data-parallel gradient bucket (devias_amd.parallel.GradSync) registers, on every parameter, the fp32 view of its flat bucket
# (`_devias_grad_out`).  When the parameter has no gradient yet, the weight-gradient kernels write straight into that view and a FRESH
# alias of it is returned to autograd: AccumulateGrad adopts a gradient tensor without copying only when nobody else holds a reference to
# that tensor object (the registered view is held by GradSync and by the parameter attribute, so returning it would be cloned -- ADVICE r2).
# A bucket is then complete the moment its last kernel finishes (no pack pass, no copy).
def _gout(p: Optional[torch.Tensor], shape=None) -> Optional[torch.Tensor]:
    if p is None or p.grad is not None:
        return None                                   # accumulation (update_freq > 1, tied uses): autograd adds a fresh tensor in place
    v = getattr(p, "_devias_grad_out", None)
    if v is None:
        return None
    if v.device != p.device:
        raise RuntimeError("gradient bucket and parameter live on different devices: build GradSync AFTER model.to(device)")
    return v.view(v.shape if shape is None else shape)


# ---- weight-gradient side stream ------------------------------------------------------------------------------------
# dX (needed by the next backward region) and dW (needed only by the optimizer) of a layer are independent: the dW GEMMs and the
# bias column sums run on a second HIP stream so their workgroups fill the tail rounds / HBM-write-bound epilogues of the dX chain
# (and vice versa).  Joined before the region returns its gradients to autograd.
import os as _os
_OVERLAP = _os.environ.get("DEVIAS_OVERLAP", "0") != "0"     # measured +0.7 % only: off by default
_SIDE = {}


def _side_stream(dev: torch.device):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


class _WgradLane:
    """`with lane.after_main():` runs the enclosed launches on the side stream once everything issued so far on the main
    stream is done; `lane.join(*outs)` makes the main stream wait and registers the outputs with it."""

    def __init__(self, dev: torch.device):
        self.on = _OVERLAP
        self.main = torch.cuda.current_stream(dev)
        self.side = _side_stream(dev) if self.on else None

    def after_main(self):
        if not self.on:
            import contextlib
            return contextlib.nullcontext()
        self.side.wait_event(self.main.record_event())
        return torch.cuda.stream(self.side)

    def join(self, *outs):
        if self.on:
            self.main.wait_stream(self.side)
            for t in outs:
                if t is not None:
                    t.record_stream(self.main)


def _f32(p: torch.Tensor) -> torch.Tensor:
    d = p.detach()
    return d if d.dtype == torch.float32 and d.is_contiguous() else d.float().contiguous()


# =====================================================================================================
# autograd Functions (each = one fused region, forward and hand-written backward over the C ABI)
# =====================================================================================================
class PatchEmbedFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pos, meta):
        ts, ps, cdt = meta
        A = ops.patch_im2col(x, ts, ps, cdt)                          # [B*N, C*ts*ps*ps]
        w = _WCACHE.get(weight, cdt)
        y = ops.gemm(A, w, bias=_f32(bias), res=pos, res_mod=pos.shape[0])
        ctx.A = A
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        weight, bias = ctx.params
        dW = ops.wgrad(dy, ctx.A, out=_gout(weight, (weight.shape[0], -1))).view(weight.shape)
        db = _take_colsum(dy, out=_gout(bias))
        ctx.A = None
        return None, dW, db, None, None


class EncoderBlockFn(Function):
    """x -> x + proj(MHSA(LN1 x)) -> + fc2(GELU(fc1(LN2 .)))   (Block.forward, modeling_slot.py:142-152; no LayerScale, drop_path 0)"""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qb, vb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, meta, ds1=None, ds2=None, drop=None):
        """drop = (E1, E2, attn): training with drop_rate / attn_drop_rate > 0 (modeling_slot.py:110,114,66).  E1 / E2: fp32 [M, D] element masks of
        proj_drop / Mlp.drop, each 0 or 1 / keep and already carrying the branch's per-sample drop_path factor (then ds1 / ds2 are None); attn =
        (keep, seed) of the softmax-matrix dropout (ops.mhsa_fwd) or None."""
        B, N, H, eps, cdt = meta
        scale = 64 ** -0.5
        E1, E2, adrop = drop if drop is not None else (None, None, None)
        assert (E1 is None) == (E2 is None) and (E1 is None or (ds1 is None and ds2 is None))
        n1w_, n1b_, n2w_, n2b_ = _f32(n1w), _f32(n1b), _f32(n2w), _f32(n2b)
        Wqkv, Wp, W1, W2 = (_WCACHE.get(w, cdt) for w in (qkvw, pw, f1w, f2w))
        u, mean1, rstd1 = ops.layernorm_fwd(x, n1w_, n1b_, eps)
        qkv_bias = torch.cat((_f32(qb), torch.zeros_like(_f32(vb)), _f32(vb)))         # modeling_slot.py:97-99
        qpre = _use_q_prescale(cdt, adrop)               # q' = q * scale * log2(e) straight from the GEMM (a weight / bias copy whose q third carries the factor): see _use_q_prescale
        if qpre:
            D3 = qkv_bias.shape[0] // 3
            qkv_bias = qkv_bias.clone()
            qkv_bias[:D3] *= _Q_PRESCALE
            qkv = ops.gemm(u, _WCACHE.get_qscaled(qkvw, D3, _Q_PRESCALE, cdt), bias=qkv_bias)
        else:
            qkv = ops.gemm(u, Wqkv, bias=qkv_bias)                                      # [M, 3D] == [B,N,3,H,64]
        o, lse = ops.mhsa_fwd(qkv, B, N, H, scale, drop=adrop, q_prescaled=qpre)
        if E1 is None:
            x1 = ops.gemm(o, Wp, bias=_f32(pb), res=x, row_scale=ds1, rows_per_scale=N)   # x + drop_path(proj(.))
        else:
            x1 = ops.mul_mask(ops.gemm(o, Wp, bias=_f32(pb)), E1, x)                      # x + drop_path(proj_drop(proj(.)))
        u2, mean2, rstd2 = ops.layernorm_fwd(x1, n2w_, n2b_, eps)
        hpre = torch.empty((x.shape[0], W1.shape[0]), dtype=cdt, device=x.device)
        hact = ops.gemm(u2, W1, bias=_f32(f1b), act=ACT_GELU, aux_out=hpre)
        if E2 is None:
            x2 = ops.gemm(hact, W2, bias=_f32(f2b), res=x1, row_scale=ds2, rows_per_scale=N)   # x1 + drop_path(mlp(.))
        else:
            x2 = ops.mul_mask(ops.gemm(hact, W2, bias=_f32(f2b)), E2, x1)                 # x1 + drop_path(drop(fc2(.)))
        ctx.meta = meta
        ctx.drop = (E1, E2, adrop)
        ctx.qpre = qpre
        ctx.ds = (ds1, ds2)
        ctx.params = (n1w, n1b, qkvw, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, qb, vb)
        ctx.saved = (x, u, mean1, rstd1, qkv, o, lse, x1, u2, mean2, rstd2, hpre, hact, n1w_, n2w_, Wqkv, Wp, W1, W2)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        B, N, H, eps, cdt = ctx.meta
        scale = 64 ** -0.5
        (x, u, mean1, rstd1, qkv, o, lse, x1, u2, mean2, rstd2, hpre, hact, n1w_, n2w_, Wqkv, Wp, W1, W2) = ctx.saved
        ctx.saved = None
        dx2 = dx2.contiguous()
        D = x.shape[1]
        dev = x.device
        ds1, ds2 = ctx.ds
        E1, E2, adrop = ctx.drop
        lane = _WgradLane(dev)
        (p_n1w, p_n1b, p_qkvw, p_pw, p_pb, p_n2w, p_n2b, p_f1w, p_f1b, p_f2w, p_f2b, p_qb, p_vb) = ctx.params
        f32 = lambda n: torch.empty((n,), dtype=torch.float32, device=dev)          # noqa: E731
        dst = lambda p_, n: _gout(p_) if _gout(p_) is not None else f32(n)          # noqa: E731  (gradient bucket view, or a fresh buffer)
        # ---- MLP branch (g2 = gradient of the branch output: dx2 scaled by the per-sample stochastic-depth factor, if any)
        if E2 is not None:
            g2 = ops.mul_mask(dx2, E2)
            db2 = ops.colsum(g2, out=_gout(p_f2b))
        elif ds2 is None:
            g2, db2 = dx2, _take_colsum(dx2, out=_gout(p_f2b))                          # fc2 bias gradient
        else:
            g2 = ops.row_scale(dx2, ds2, N)
            db2 = ops.colsum(g2, out=_gout(p_f2b))
        with lane.after_main():
            dW2 = ops.wgrad(g2, hact, out=_gout(p_f2w))
        db1 = dst(p_f1b, W1.shape[0])
        dhpre = ops.gemm(g2, W2, trans_b=True, act=ACT_DGELU, aux_in=hpre, colsum=db1)   # (g2 W2) * gelu'(pre); db1 = colsum
        with lane.after_main():
            dW1 = ops.wgrad(dhpre, u2, out=_gout(p_f1w))
        du2 = ops.gemm(dhpre, W1, trans_b=True)
        dbp = dst(p_pb, D)
        dx1, dn2w, dn2b = ops.layernorm_bwd(du2, x1, n2w_, mean2, rstd2, dres=dx2, dx_colsum=dbp,
                                            dgamma=_gout(p_n2w), dbeta=_gout(p_n2b))      # + residual gradient; dbp = colsum(dx1)
        # ---- attention branch
        if E1 is not None:
            g1 = ops.mul_mask(dx1, E1)
            dbp = ops.colsum(g1, out=dbp)
        elif ds1 is None:
            g1 = dx1
        else:
            g1 = ops.row_scale(dx1, ds1, N)
            dbp = ops.colsum(g1, out=dbp)
        with lane.after_main():
            dWp = ops.wgrad(g1, o, out=_gout(p_pw))
        dbq, dbv = dst(p_qb, D), dst(p_vb, D)                                  # q_bias | (k: no bias) | v_bias, each to its own destination
        if ops.mhsa_bwd_dv_from_do(g1.dtype, adrop):
            # softmax rows sum to one: sum_keys dV = sum_queries dO -- the v_bias gradient is the column sum of d_o, taken from the GEMM that produces it
            d_o = ops.gemm(g1, Wp, trans_b=True, colsum=dbv)
            dqkv = ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, scale, drop=adrop, bias_out=(dbq, None), q_prescaled=ctx.qpre)
        else:
            d_o = ops.gemm(g1, Wp, trans_b=True)
            dqkv = ops.mhsa_bwd(qkv, o, d_o, lse, B, N, H, scale, drop=adrop, bias_out=(dbq, dbv), q_prescaled=ctx.qpre)
        with lane.after_main():
            dWqkv = ops.wgrad(dqkv, u, out=_gout(p_qkvw))
        du = ops.gemm(dqkv, Wqkv, trans_b=True)
        dxs = f32(D)
        dx, dn1w, dn1b = ops.layernorm_bwd(du, x, n1w_, mean1, rstd1, dres=dx1, dx_colsum=dxs, dgamma=_gout(p_n1w), dbeta=_gout(p_n1b))
        _publish_colsum(dx, dxs)
        lane.join(dW2, dW1, dWp, dWqkv, dbq, dbv)
        return (dx, dn1w, dn1b, dWqkv, dbq, dbv, dWp, dbp, dn2w, dn2b, dW1, db1, dW2, db2, None, None, None, None)


class DropMaskFn(Function):
    """y = x * mask with a caller-drawn fp32 mask of 0 / (1 / keep): nn.Dropout with its mask made explicit (pos_drop, modeling_slot.py:280,356)"""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.mask = mask
        return ops.mul_mask(x, mask)

    @staticmethod
    def backward(ctx, dy):
        return ops.mul_mask(dy.contiguous(), ctx.mask), None


class DropoutSource:
    """Where the training-time dropout masks of the encoder come from.  The default draws them from torch's generators, as nn.Dropout does
    (element masks: the device generator, like the drop_path masks; the seed of the attention-matrix mask: the CPU generator, no device sync).
    Tests and tests/golden/make_goldens.py replace it (VisionTransformer.dropout_source) to give the reference, the oracle and the kernels the SAME masks."""

    def element_mask(self, kind: str, block: int, shape, keep: float, device) -> torch.Tensor:
        """fp32 mask of 0 / (1 / keep); kind in {'pos', 'proj', 'mlp'}"""
        return ((keep + torch.rand(shape, device=device, dtype=torch.float32)).floor() / keep).contiguous()

    def path_scale(self, block: int, B: int, keep: float, device) -> torch.Tensor:
        """fp32 [2, B]: timm 0.4.12 drop_path (modeling_slot.py:36-47) masks of the attention and the MLP branch, 0 / (1 / keep) per sample"""
        return ((keep + torch.rand((2, B), device=device, dtype=torch.float32)).floor() / keep).contiguous()

    def attn_seed(self, block: int) -> int:
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


_LAYER_KEYS = ("to_q", "to_k", "to_v", "to_out_w", "to_out_b", "norm_w", "norm_b", "ctx_w", "ctx_b",
               "ff0_w", "ff0_b", "ff3_w", "ff3_b", "ffn_w", "ffn_b")


class AggBlockFn(Function):
    """encoder output -> final LN -> AggregationBlock -> (slots [B*S, D], attn [B*h, S, N] fp32)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, latents, last_w, last_b, meta, *layer_params):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = meta
        nset = 1 if tied else depth
        assert len(layer_params) == nset * len(_LAYER_KEYS)
        LP = [dict(zip(_LAYER_KEYS, layer_params[i * 15:(i + 1) * 15])) for i in range(nset)]
        inner = heads * dh
        scale = dh ** -0.5
        dev = x.device
        M = x.shape[0]
        feats, m0, r0 = ops.layernorm_fwd(x, _f32(norm_w), _f32(norm_b), eps_enc)        # modeling_slot.py:373
        # context LayerNorm + fused K|V projection, once per distinct weight set (the reference redoes it every layer)
        kvsets = []
        for P in LP:
            c, mc, rc = ops.layernorm_fwd(feats, _f32(P["ctx_w"]), _f32(P["ctx_b"]), eps_agg)
            Wkv = _WCACHE.get_cat((P["to_k"], P["to_v"]), cdt)
            kv = ops.gemm(c, Wkv)                                                         # [M, 2*inner] == [B,N,2,h,dh]
            kvsets.append((c, mc, rc, Wkv, kv))
        xs = ops.rows_broadcast(_f32(latents), B * S, cdt)                                # agg_block.py:112-114
        q_stack = torch.empty((depth, B * S, inner), dtype=cdt, device=dev)
        attn_stack = torch.empty((depth, B * heads, S, N), dtype=torch.float32, device=dev)
        rsum_stack = torch.empty((depth, B * heads, S), dtype=torch.float32, device=dev)
        layers = []
        for l in range(depth):
            P = LP[0 if tied else l]
            kv = kvsets[0 if tied else l][4]
            Wq, Wo, W1, W2 = (_WCACHE.get(P[k], cdt) for k in ("to_q", "to_out_w", "ff0_w", "ff3_w"))
            qn, mq, rq = ops.layernorm_fwd(xs, _f32(P["norm_w"]), _f32(P["norm_b"]), eps_agg)
            q = ops.gemm(qn, Wq, out=q_stack[l])
            _, _, o = ops.slot_attn_fwd(q, kv, B, S, N, heads, dh, scale, attn_out=attn_stack[l], rsum_out=rsum_stack[l])
            xs1 = ops.gemm(o, Wo, bias=_f32(P["to_out_b"]), res=xs)
            f, mf, rf = ops.layernorm_fwd(xs1, _f32(P["ffn_w"]), _f32(P["ffn_b"]), eps_agg)
            fpre = torch.empty((B * S, W1.shape[0]), dtype=cdt, device=dev)
            fact = ops.gemm(f, W1, bias=_f32(P["ff0_b"]), act=ACT_GELU, aux_out=fpre)
            xs2 = ops.gemm(fact, W2, bias=_f32(P["ff3_b"]), res=xs1)
            layers.append((xs, mq, rq, qn, o, xs1, mf, rf, f, fpre, fact, Wq, Wo, W1, W2))
            xs = xs2
        slots, ml, rl = ops.layernorm_fwd(xs, _f32(last_w), _f32(last_b), eps_agg)
        ctx.meta = meta
        ctx.saved = (x, m0, r0, feats, kvsets, layers, q_stack, attn_stack, rsum_stack, xs, ml, rl,
                     _f32(norm_w), _f32(last_w), [{k: _f32(P[k]) for k in ("norm_w", "ctx_w", "ffn_w")} for P in LP])
        return slots, attn_stack[depth - 1]

    @staticmethod
    def backward(ctx, dslots, dattn):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = ctx.meta
        (x, m0, r0, feats, kvsets, layers, q_stack, attn_stack, rsum_stack, xs_last, ml, rl, norm_w, last_w, LNW) = ctx.saved
        ctx.saved = None
        nset = 1 if tied else depth
        inner = heads * dh
        scale = dh ** -0.5
        dev = x.device
        G = [dict() for _ in range(nset)]        # gradient accumulators per distinct weight set

        def acc_w(si, key, dY, X):
            if key in G[si]:
                ops.wgrad(dY, X, out=G[si][key], beta=1.0)
            else:
                G[si][key] = ops.wgrad(dY, X)

        def acc_b(si, key, dY):
            if key in G[si]:
                ops.colsum(dY, out=G[si][key], beta=1.0)
            else:
                G[si][key] = ops.colsum(dY)

        def ln_bwd(si, kw, kb, dy, xin, gamma, mean, rstd, dres):
            if kw in G[si]:
                dxo, _, _ = ops.layernorm_bwd(dy, xin, gamma, mean, rstd, dres=dres, dgamma=G[si][kw], dbeta=G[si][kb], beta_acc=1.0)
            else:
                dxo, G[si][kw], G[si][kb] = ops.layernorm_bwd(dy, xin, gamma, mean, rstd, dres=dres)
            return dxo

        dxs, dlast_w, dlast_b = ops.layernorm_bwd(dslots.contiguous(), xs_last, last_w, ml, rl)
        do_stack = torch.empty((depth, B * S, inner), dtype=cdt, device=dev)
        ds_stack = torch.empty((depth, B * heads, S, N), dtype=torch.float32, device=dev)
        dattn_ext = dattn.contiguous() if dattn is not None else None
        for l in reversed(range(depth)):
            si = 0 if tied else l
            (xs_in, mq, rq, qn, o, xs1, mf, rf, f, fpre, fact, Wq, Wo, W1, W2) = layers[l]
            kv = kvsets[si][4]
            # feed-forward: xs2 = xs1 + W2 gelu(W1 LN(xs1) + b1) + b2
            dfpre = ops.gemm(dxs, W2, trans_b=True, act=ACT_DGELU, aux_in=fpre)
            acc_w(si, "ff3_w", dxs, fact); acc_b(si, "ff3_b", dxs)
            df = ops.gemm(dfpre, W1, trans_b=True)
            acc_w(si, "ff0_w", dfpre, f); acc_b(si, "ff0_b", dfpre)
            dxs1 = ln_bwd(si, "ffn_w", "ffn_b", df, xs1, LNW[si]["ffn_w"], mf, rf, dxs)
            # cross attention: xs1 = xs + Wo o + bo
            d_o = ops.gemm(dxs1, Wo, trans_b=True, out=do_stack[l])
            acc_w(si, "to_out_w", dxs1, o); acc_b(si, "to_out_b", dxs1)
            dq, _ = ops.slot_attn_bwd(q_stack[l], kv, attn_stack[l], rsum_stack[l], o, d_o,
                                      dattn_ext if l == depth - 1 else None, B, S, N, heads, dh, scale, ds_out=ds_stack[l])
            dqn = ops.gemm(dq, Wq, trans_b=True)
            acc_w(si, "to_q", dq, qn)
            dxs = ln_bwd(si, "norm_w", "norm_b", dqn, xs_in, LNW[si]["norm_w"], mq, rq, dxs1)
        dlatents = ops.rows_reduce_mod(dxs, S)
        # deferred K/V gradients: one pass per distinct K/V over all the layers that used it
        dfeats = None
        for si in range(nset):
            c, mc, rc, Wkv, kv = kvsets[si]
            if tied:
                dkv = ops.slot_attn_kv_grad(q_stack, do_stack, ds_stack, attn_stack, rsum_stack, depth, B, S, N, heads, dh, scale)
            else:
                dkv = ops.slot_attn_kv_grad(q_stack[si:si + 1], do_stack[si:si + 1], ds_stack[si:si + 1], attn_stack[si:si + 1],
                                            rsum_stack[si:si + 1], 1, B, S, N, heads, dh, scale)
            dc = ops.gemm(dkv, Wkv, trans_b=True)
            dWkv = ops.wgrad(dkv, c)
            G[si]["to_k"], G[si]["to_v"] = dWkv[:inner], dWkv[inner:]
            dfeats_i, G[si]["ctx_w"], G[si]["ctx_b"] = ops.layernorm_bwd(dc, feats, LNW[si]["ctx_w"], mc, rc, dres=dfeats)
            dfeats = dfeats_i
        dxs = torch.empty((x.shape[1],), dtype=torch.float32, device=dev)
        dx, dnorm_w, dnorm_b = ops.layernorm_bwd(dfeats, x, norm_w, m0, r0, dx_colsum=dxs)
        _publish_colsum(dx, dxs)
        flat = []
        for si in range(nset):
            flat += [G[si][k] for k in _LAYER_KEYS]
        return (dx, dnorm_w, dnorm_b, dlatents, dlast_w, dlast_b, None, *flat)


# ---- folded aggregation block -------------------------------------------------------------------------------------------------------------
# sim = scale (Wk_h^T q) . c_j and o = Wv_h sum_j Abar c_j: the to_k / to_v projections move to the slot side (composite D x D weights per head,
# built once per forward), the per-layer stream is the context c [M, D] instead of K|V [M, 2*h*512], and the K|V GEMM + its dgrad + wgrad vanish
# (csrc/slot_attn.hip "folded form").  Same parameters, same outputs, same gradients; summation order differs (fp32 round-off).
_AGG_FOLD = _os.environ.get("DEVIAS_AGG_FOLD", "1") != "0"


def _composites(P, heads, dh, D, cdt):
    """Wqk [h*D, D] (Wqk_h = Wk_h^T Wq_h) and Wov [D, h*D] (Wov_h = Wo_h Wv_h) in the compute dtype, + the operands for their backward"""
    Wq, Wk, Wv, Wo = (_WCACHE.get(P[k], cdt) for k in ("to_q", "to_k", "to_v", "to_out_w"))
    inner = heads * dh
    Wqk = torch.empty((heads * D, D), dtype=cdt, device=Wq.device)
    ops.gemm_batched(Wk, Wq, Wqk, D, D, dh, lda=D, ldb=D, ldc=D, stride_a=dh * D, stride_b=dh * D, stride_c=D * D, batch=heads,
                     trans_a=True, trans_b=True)
    Wov = torch.empty((D, heads * D), dtype=cdt, device=Wq.device)
    ops.gemm_batched(Wo, Wv, Wov, D, D, dh, lda=inner, ldb=D, ldc=heads * D, stride_a=dh, stride_b=dh * D, stride_c=D, batch=heads,
                     trans_b=True)
    return Wqk, Wov, (Wq, Wk, Wv, Wo)


def _composite_grads(dWqk, dWov, W4, heads, dh, D, cdt):
    """gradients of to_q / to_k / to_v / to_out.weight (fp32) from those of the composites (fp32 accumulators over the layers)"""
    Wq, Wk, Wv, Wo = W4
    inner = heads * dh
    dev = Wq.device
    gqk = dWqk if cdt == torch.float32 else ops.cast(dWqk, cdt)
    gov = dWov if cdt == torch.float32 else ops.cast(dWov, cdt)
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)          # noqa: E731
    dWq, dWk, dWv, dWo = f(inner, D), f(inner, D), f(inner, D), f(D, inner)
    ops.gemm_batched(Wk, gqk, dWq, dh, D, D, lda=D, ldb=D, ldc=D, stride_a=dh * D, stride_b=D * D, stride_c=dh * D, batch=heads, trans_b=True)
    ops.gemm_batched(Wq, gqk, dWk, dh, D, D, lda=D, ldb=D, ldc=D, stride_a=dh * D, stride_b=D * D, stride_c=dh * D, batch=heads)
    ops.gemm_batched(gov, Wv, dWo, D, dh, D, lda=heads * D, ldb=D, ldc=inner, stride_a=D, stride_b=dh * D, stride_c=dh, batch=heads)
    ops.gemm_batched(Wo, gov, dWv, dh, D, D, lda=inner, ldb=heads * D, ldc=D, stride_a=dh, stride_b=D, stride_c=dh * D, batch=heads,
                     trans_a=True, trans_b=True)
    return dWq, dWk, dWv, dWo


class AggBlockFoldFn(Function):
    """encoder output -> final LN -> AggregationBlock (folded slot attention) -> (slots [B*S, D], attn [B*h, S, N] fp32)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, latents, last_w, last_b, meta, *layer_params):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = meta
        nset = 1 if tied else depth
        assert len(layer_params) == nset * len(_LAYER_KEYS)
        LP = [dict(zip(_LAYER_KEYS, layer_params[i * 15:(i + 1) * 15])) for i in range(nset)]
        scale = dh ** -0.5
        dev = x.device
        D = x.shape[1]
        feats, m0, r0 = ops.layernorm_fwd(x, _f32(norm_w), _f32(norm_b), eps_enc)        # modeling_slot.py:373
        sets = []
        for P in LP:                                                                      # context LayerNorm + composite weights, once per weight set
            c, mc, rc = ops.layernorm_fwd(feats, _f32(P["ctx_w"]), _f32(P["ctx_b"]), eps_agg)
            Wqk, Wov, W4 = _composites(P, heads, dh, D, cdt)
            sets.append((c, mc, rc, Wqk, Wov, W4))
        xs = ops.rows_broadcast(_f32(latents), B * S, cdt)                                # agg_block.py:112-114
        qp_stack = torch.empty((depth, B * S, heads * D), dtype=cdt, device=dev)
        attn_stack = torch.empty((depth, B * heads, S, N), dtype=torch.float32, device=dev)
        rsum_stack = torch.empty((depth, B * heads, S), dtype=torch.float32, device=dev)
        layers = []
        for l in range(depth):
            si = 0 if tied else l
            P = LP[si]
            c, _, _, Wqk, Wov, _ = sets[si]
            W1, W2 = (_WCACHE.get(P[k], cdt) for k in ("ff0_w", "ff3_w"))
            qn, mq, rq = ops.layernorm_fwd(xs, _f32(P["norm_w"]), _f32(P["norm_b"]), eps_agg)
            qp = ops.gemm(qn, Wqk, out=qp_stack[l])
            _, _, z = ops.slotf_fwd(qp, c, B, S, N, heads, D, scale, attn_out=attn_stack[l], rsum_out=rsum_stack[l])
            xs1 = ops.gemm(z, Wov, bias=_f32(P["to_out_b"]), res=xs)
            f, mf, rf = ops.layernorm_fwd(xs1, _f32(P["ffn_w"]), _f32(P["ffn_b"]), eps_agg)
            fpre = torch.empty((B * S, W1.shape[0]), dtype=cdt, device=dev)
            fact = ops.gemm(f, W1, bias=_f32(P["ff0_b"]), act=ACT_GELU, aux_out=fpre)
            xs2 = ops.gemm(fact, W2, bias=_f32(P["ff3_b"]), res=xs1)
            layers.append((xs, mq, rq, qn, z, xs1, mf, rf, f, fpre, fact, W1, W2))
            xs = xs2
        slots, ml, rl = ops.layernorm_fwd(xs, _f32(last_w), _f32(last_b), eps_agg)
        ctx.meta = meta
        ctx.saved = (x, m0, r0, feats, sets, layers, qp_stack, attn_stack, rsum_stack, xs, ml, rl,
                     _f32(norm_w), _f32(last_w), [{k: _f32(P[k]) for k in ("norm_w", "ctx_w", "ffn_w")} for P in LP])
        return slots, attn_stack[depth - 1]

    @staticmethod
    def backward(ctx, dslots, dattn):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = ctx.meta
        (x, m0, r0, feats, sets, layers, qp_stack, attn_stack, rsum_stack, xs_last, ml, rl, norm_w, last_w, LNW) = ctx.saved
        ctx.saved = None
        nset = 1 if tied else depth
        scale = dh ** -0.5
        dev = x.device
        D = x.shape[1]
        G = [dict() for _ in range(nset)]        # gradient accumulators per distinct weight set

        def acc_w(si, key, dY, X):
            if key in G[si]:
                ops.wgrad(dY, X, out=G[si][key], beta=1.0)
            else:
                G[si][key] = ops.wgrad(dY, X)

        def acc_b(si, key, dY):
            if key in G[si]:
                ops.colsum(dY, out=G[si][key], beta=1.0)
            else:
                G[si][key] = ops.colsum(dY)

        def ln_bwd(si, kw, kb, dy, xin, gamma, mean, rstd, dres):
            if kw in G[si]:
                dxo, _, _ = ops.layernorm_bwd(dy, xin, gamma, mean, rstd, dres=dres, dgamma=G[si][kw], dbeta=G[si][kb], beta_acc=1.0)
            else:
                dxo, G[si][kw], G[si][kb] = ops.layernorm_bwd(dy, xin, gamma, mean, rstd, dres=dres)
            return dxo

        dxs, dlast_w, dlast_b = ops.layernorm_bwd(dslots.contiguous(), xs_last, last_w, ml, rl)
        dz_stack = torch.empty((depth, B * S, heads * D), dtype=cdt, device=dev)
        ds_stack = torch.empty((depth, B * heads, S, N), dtype=torch.float32, device=dev)
        dattn_ext = dattn.contiguous() if dattn is not None else None
        for l in reversed(range(depth)):
            si = 0 if tied else l
            (xs_in, mq, rq, qn, z, xs1, mf, rf, f, fpre, fact, W1, W2) = layers[l]
            c, _, _, Wqk, Wov, _ = sets[si]
            # feed-forward: xs2 = xs1 + W2 gelu(W1 LN(xs1) + b1) + b2
            dfpre = ops.gemm(dxs, W2, trans_b=True, act=ACT_DGELU, aux_in=fpre)
            acc_w(si, "ff3_w", dxs, fact); acc_b(si, "ff3_b", dxs)
            df = ops.gemm(dfpre, W1, trans_b=True)
            acc_w(si, "ff0_w", dfpre, f); acc_b(si, "ff0_b", dfpre)
            dxs1 = ln_bwd(si, "ffn_w", "ffn_b", df, xs1, LNW[si]["ffn_w"], mf, rf, dxs)
            # cross attention: xs1 = xs + Wov z + bo
            dz = ops.gemm(dxs1, Wov, trans_b=True, out=dz_stack[l])
            acc_w(si, "Wov", dxs1, z); acc_b(si, "to_out_b", dxs1)
            dqp, _ = ops.slotf_bwd(c, attn_stack[l], rsum_stack[l], z, dz, dattn_ext if l == depth - 1 else None,
                                   B, S, N, heads, D, scale, ds_out=ds_stack[l])
            dqn = ops.gemm(dqp, Wqk, trans_b=True)
            acc_w(si, "Wqk", dqp, qn)
            dxs = ln_bwd(si, "norm_w", "norm_b", dqn, xs_in, LNW[si]["norm_w"], mq, rq, dxs1)
        dlatents = ops.rows_reduce_mod(dxs, S)
        # deferred context gradient: one pass per distinct context over all the layers that used it; composite -> parameter gradients
        dfeats = None
        for si in range(nset):
            c, mc, rc, Wqk, Wov, W4 = sets[si]
            sl = slice(0, depth) if tied else slice(si, si + 1)
            nl = depth if tied else 1
            dc = ops.slotf_context_grad(attn_stack[sl], rsum_stack[sl], ds_stack[sl], dz_stack[sl], qp_stack[sl], nl, B, S, N, heads, D, scale)
            G[si]["to_q"], G[si]["to_k"], G[si]["to_v"], G[si]["to_out_w"] = _composite_grads(G[si].pop("Wqk"), G[si].pop("Wov"), W4, heads, dh, D, cdt)
            dfeats_i, G[si]["ctx_w"], G[si]["ctx_b"] = ops.layernorm_bwd(dc, feats, LNW[si]["ctx_w"], mc, rc, dres=dfeats)
            dfeats = dfeats_i
        dxs = torch.empty((x.shape[1],), dtype=torch.float32, device=dev)
        dx, dnorm_w, dnorm_b = ops.layernorm_bwd(dfeats, x, norm_w, m0, r0, dx_colsum=dxs)
        _publish_colsum(dx, dxs)
        flat = []
        for si in range(nset):
            flat += [G[si][k] for k in _LAYER_KEYS]
        return (dx, dnorm_w, dnorm_b, dlatents, dlast_w, dlast_b, None, *flat)


class HeadFn(Function):
    """slots -> (slots_head = head(slots), mask_predictions = MaskPredictor(slots))  (modeling_slot.py:392-393, 209-216)"""

    @staticmethod
    def forward(ctx, slots, hw, hb, w0, b0, w2, b2, w4, b4, cdt):
        Wh, W0, W2, W4 = (_WCACHE.get(w, cdt) for w in (hw, w0, w2, w4))
        Z = ops.gemm(slots, Wh, bias=_f32(hb))
        m1 = ops.gemm(slots, W0, bias=_f32(b0), act=ACT_RELU)
        m2 = ops.gemm(m1, W2, bias=_f32(b2), act=ACT_RELU)
        Mk = ops.gemm(m2, W4, bias=_f32(b4), act=ACT_SIGMOID)
        ctx.saved = (slots, m1, m2, Mk, Wh, W0, W2, W4)
        return Z, Mk

    @staticmethod
    def backward(ctx, dZ, dM):
        slots, m1, m2, Mk, Wh, W0, W2, W4 = ctx.saved
        ctx.saved = None
        dZ = dZ.contiguous() if dZ is not None else torch.zeros((slots.shape[0], Wh.shape[0]), dtype=slots.dtype, device=slots.device)
        dM = dM.contiguous() if dM is not None else torch.zeros_like(Mk)
        dp3 = ops.act_bwd(dM, Mk, ACT_SIGMOID)
        dW4, db4 = ops.wgrad(dp3, m2), ops.colsum(dp3)
        dp2 = ops.gemm(dp3, W4, trans_b=True, act=ACT_DRELU, aux_in=m2)
        dW2, db2 = ops.wgrad(dp2, m1), ops.colsum(dp2)
        dp1 = ops.gemm(dp2, W2, trans_b=True, act=ACT_DRELU, aux_in=m1)
        dW0, db0 = ops.wgrad(dp1, slots), ops.colsum(dp1)
        ds_m = ops.gemm(dp1, W0, trans_b=True)
        dslots = ops.gemm(dZ, Wh, trans_b=True, res=ds_m)
        dWh, dbh = ops.wgrad(dZ, slots), ops.colsum(dZ)
        return dslots, dWh, dbh, dW0, db0, dW2, db2, dW4, db4, None


class HeadMlpFn(Function):
    """head_type='mlp' (MLPHead, modeling_slot.py:23-34, 307-313): slots -> (slots_head = fc2(relu(fc1(dropout(slots)))), mask_predictions = MaskPredictor(slots)).
    Not on any DEVIAS recipe's path (docs/TRAIN.md uses 'linear'), so it is composed from the per-kernel calls (GEMM with the ReLU / dReLU / Sigmoid
    epilogues, weight gradients, column sums) instead of having a fused region of its own."""

    @staticmethod
    def forward(ctx, slots, f1w, f1b, f2w, f2b, w0, b0, w2, b2, w4, b4, cdt, drop_mask=None):
        F1, F2, W0, W2, W4 = (_WCACHE.get(w, cdt) for w in (f1w, f2w, w0, w2, w4))
        slots = slots.contiguous()
        xin = ops.mul_mask(slots, drop_mask) if drop_mask is not None else slots          # fc_dropout applies to the head's input only (:393)
        t = ops.gemm(xin, F1, bias=_f32(f1b), act=ACT_RELU)
        Z = ops.gemm(t, F2, bias=_f32(f2b))
        m1 = ops.gemm(slots, W0, bias=_f32(b0), act=ACT_RELU)
        m2 = ops.gemm(m1, W2, bias=_f32(b2), act=ACT_RELU)
        Mk = ops.gemm(m2, W4, bias=_f32(b4), act=ACT_SIGMOID)
        ctx.saved = (slots, xin, t, m1, m2, Mk, F1, F2, W0, W2, W4, drop_mask)
        return Z, Mk

    @staticmethod
    def backward(ctx, dZ, dM):
        if ctx.saved is None:
            raise RuntimeError("HeadMlpFn: backward already consumed the saved activations (one backward per forward)")
        slots, xin, t, m1, m2, Mk, F1, F2, W0, W2, W4, drop_mask = ctx.saved
        ctx.saved = None
        dZ = dZ.contiguous() if dZ is not None else torch.zeros((slots.shape[0], F2.shape[0]), dtype=slots.dtype, device=slots.device)
        dM = dM.contiguous() if dM is not None else torch.zeros_like(Mk)
        dp3 = ops.act_bwd(dM, Mk, ACT_SIGMOID)
        dW4, db4 = ops.wgrad(dp3, m2), ops.colsum(dp3)
        dp2 = ops.gemm(dp3, W4, trans_b=True, act=ACT_DRELU, aux_in=m2)
        dW2, db2 = ops.wgrad(dp2, m1), ops.colsum(dp2)
        dp1 = ops.gemm(dp2, W2, trans_b=True, act=ACT_DRELU, aux_in=m1)
        dW0, db0 = ops.wgrad(dp1, slots), ops.colsum(dp1)
        ds_m = ops.gemm(dp1, W0, trans_b=True)
        dF2, dbf2 = ops.wgrad(dZ, t), ops.colsum(dZ)
        dt = ops.gemm(dZ, F2, trans_b=True, act=ACT_DRELU, aux_in=t)
        dF1, dbf1 = ops.wgrad(dt, xin), ops.colsum(dt)
        if drop_mask is not None:
            dslots = ops.mul_mask(ops.gemm(dt, F1, trans_b=True), drop_mask, ds_m)
        else:
            dslots = ops.gemm(dt, F1, trans_b=True, res=ds_m)
        return dslots, dF1, dbf1, dF2, dbf2, dW0, db0, dW2, db2, dW4, db4, None, None


# =====================================================================================================
# fused regions: ONE library call per region and direction (devias_encoder_block_* / devias_agg_block_* / devias_head_*)
# =====================================================================================================
# The kernel sequence of a region is issued by the library (csrc/regions.hip) -- the same launches in the same order as the per-kernel
# Functions above, bitwise the same results (tests/test_regions_gpu.py) -- so a step costs ~40 Python -> library hops instead of ~750.
# What backward needs lives in ONE arena tensor per region call, backward temporaries in a grow-only scratch buffer per stream, the
# parameter gradients of a region in one flat fp32 tensor (or straight in the data-parallel gradient bucket views).
import ctypes as _ct

from . import _lib as _L

_REGIONS = _os.environ.get("DEVIAS_REGIONS", "1") != "0"
_scratch_bufs = {}


def _scratch(nbytes: int, device) -> torch.Tensor:
    """grow-only backward scratch per (device, stream): the temporaries of a region's backward are dead when its kernels have run, and the
    kernels of one stream are ordered"""
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch_bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _scratch_bufs.pop(key, None)
        buf = _scratch_bufs[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return buf


class _GradDest:
    """fp32 gradient destinations of one region: a parameter whose gradient bucket view is free gets that view (a fresh alias, see _gout);
    the others are carved out of ONE flat tensor"""

    def __init__(self, items, device):
        self.out = []
        views = [(_gout(p, shape) if p is not None else None) for p, shape in items]
        total = 0
        for (p, shape), v in zip(items, views):
            if v is None:
                total += (math.prod(shape) + 63) // 64 * 64           # 256-byte aligned pieces
        flat = torch.empty((total,), dtype=torch.float32, device=device) if total else None
        off = 0
        for (p, shape), v in zip(items, views):
            if v is None:
                n = math.prod(shape)
                v = flat[off:off + n].view(shape)
                off += (n + 63) // 64 * 64
            self.out.append(v)

    def ptrs(self):
        return [t.data_ptr() for t in self.out]


def _qkv_bias(qb: torch.Tensor, vb: torch.Tensor) -> torch.Tensor:
    """fp32 [3D] = q_bias | zeros | v_bias (modeling_slot.py:97-99), rebuilt only when one of the two changed"""
    stamp = (_WCACHE._stamp(qb), _WCACHE._stamp(vb))
    hit = _WCACHE._qkvb.get(qb)
    if hit is not None and hit[0] == stamp and hit[1]() is vb:
        return hit[2]
    t = torch.cat((_f32(qb), torch.zeros_like(_f32(vb)), _f32(vb))).contiguous()
    _WCACHE._qkvb[qb] = (stamp, weakref.ref(vb), t)
    return t


# scale * log2(e) for head dim 64: the factor the q third of the qkv projection carries where the attention kernels run with DEVIAS_ATTN_Q_PRESCALED
_Q_PRESCALE = (64 ** -0.5) * 1.4426950408889634


def _use_q_prescale(cdt, drop=None) -> bool:
    """bf16 encoder blocks without attention dropout: the qkv GEMM runs on a copy of its weight / bias whose q third carries scale * log2(e), so that forward, dQ and
    dK / dV kernels multiply the same bf16 operands (library option attn_qpre, default 1; 0 = every kernel scales on its own: A/B aid)."""
    return cdt == torch.bfloat16 and drop is None and ops.get_option("attn_qpre") != 0


class EncoderBlockRegionFn(Function):
    """EncoderBlockFn as one devias_encoder_block_fwd / _bwd call (Block.forward, modeling_slot.py:142-152)"""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qb, vb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, meta, ds1=None, ds2=None):
        B, N, H, eps, cdt = meta
        lib = _L.load()
        dev = x.device
        D = x.shape[1]
        hid = f1w.shape[0]
        dt = ops.dt_code(cdt)
        x = x.contiguous()
        keep = [_f32(n1w), _f32(n1b), _f32(n2w), _f32(n2b)] + [_WCACHE.get(w, cdt) for w in (qkvw, pw, f1w, f2w)] + \
               [_qkv_bias(qb, vb), _f32(pb), _f32(f1b), _f32(f2b)]
        save = torch.empty((lib.devias_encoder_block_save_bytes(B, N, D, H, hid, dt),), dtype=torch.uint8, device=dev)
        ws = ops.workspace(lib.devias_encoder_block_workspace_bytes(B, N, D, H, hid, dt), dev)
        a = _L.BlockArgs()
        a.B, a.N, a.D, a.H, a.hidden, a.dtype, a.eps = B, N, D, H, hid, dt, eps
        (a.n1w, a.n1b, a.n2w, a.n2b, a.Wqkv, a.Wp, a.W1, a.W2, a.qkv_bias, a.pb, a.b1, a.b2) = [t.data_ptr() for t in keep]
        if _use_q_prescale(cdt):
            qs = [_WCACHE.get_qscaled(qkvw, D, _Q_PRESCALE, cdt), _WCACHE.get_qscaled(keep[8], D, _Q_PRESCALE, torch.float32)]
            keep = keep + qs
            a.WqkvS, a.qkv_biasS = qs[0].data_ptr(), qs[1].data_ptr()
        if cdt != torch.float32 and any(ctx.needs_input_grad):
            # transposed weight copies for the four dgrad GEMMs of the backward (bf16 mode: the fp32 parity kernels stage through registers and do not care)
            wt = [_WCACHE.get_t(w, cdt) for w in (qkvw, pw, f1w, f2w)]
            keep = keep + wt
            (a.WqkvT, a.WpT, a.W1T, a.W2T) = [t.data_ptr() for t in wt]
        a.ds1 = ds1.data_ptr() if ds1 is not None else None
        a.ds2 = ds2.data_ptr() if ds2 is not None else None
        a.save, a.ws, a.ws_bytes = save.data_ptr(), ws.data_ptr(), ws.numel() * 4
        x2 = torch.empty_like(x)
        _L.check(lib.devias_encoder_block_fwd(_ct.byref(a), x.data_ptr(), x2.data_ptr(), ops._stream()), "devias_encoder_block_fwd")
        ctx.args = a
        ctx.meta = meta
        ctx.keep = (keep, save, ds1, ds2, x)
        ctx.x_version = x._version
        ctx.params = (n1w, n1b, qkvw, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, qb, vb)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        B, N, H, eps, cdt = ctx.meta
        lib = _L.load()
        _region_state(ctx, "EncoderBlockRegionFn", ctx.keep[4] if ctx.keep is not None else None)
        a = ctx.args
        keep, save, ds1, ds2, x = ctx.keep
        (p_n1w, p_n1b, p_qkvw, p_pw, p_pb, p_n2w, p_n2b, p_f1w, p_f1b, p_f2w, p_f2b, p_qb, p_vb) = ctx.params
        dev = x.device
        D, hid = a.D, a.hidden
        dx2 = dx2.contiguous()
        ready = _peek_colsum(dx2) if ds2 is None else None          # colsum(dx2) = fc2 bias gradient, published by the next block's LayerNorm backward
        b2_dst = _gout(p_f2b) if ready is not None else None
        # (q_bias and v_bias are two parameters: their gradients go to two destinations -- bucket views under GradSync, else pieces of the flat tensor --
        # instead of one [3D] vector that autograd would have to slice and GradSync to copy: 24 device-to-device copies per step, +1.0 ms)
        gd = _GradDest([(p_n1w, (D,)), (p_n1b, (D,)), (p_qkvw, (3 * D, D)), (p_qb, (D,)), (p_pw, (D, D)), (p_pb, (D,)), (p_n2w, (D,)), (p_n2b, (D,)),
                        (p_f1w, (hid, D)), (p_f1b, (hid,)), (p_f2w, (D, hid)), (None if ready is not None else p_f2b, (1,) if ready is not None else (D,)),
                        (None, (D,)), (p_vb, (D,))], dev)
        g = _L.BlockGrads()
        (g.dn1w, g.dn1b, g.dWqkv, g.dbq, g.dWp, g.dbp, g.dn2w, g.dn2b, g.dW1, g.db1, g.dW2, g.db2, g.dx_colsum, g.dbv) = gd.ptrs()
        g.db2_done = 1 if ready is not None else 0
        ws = ops.workspace(a.ws_bytes, dev)                 # (the stream's workspace may have been re-allocated larger since forward)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        nscr = lib.devias_encoder_block_scratch_bytes(B, N, D, H, hid, a.dtype)
        scr = _scratch(nscr, dev)
        dx = torch.empty_like(x)
        _L.check(lib.devias_encoder_block_bwd(_ct.byref(a), x.data_ptr(), dx2.data_ptr(), dx.data_ptr(), _ct.byref(g), scr.data_ptr(), scr.numel(), ops._stream()),
                 "devias_encoder_block_bwd")
        ctx.keep = ctx.args = None
        (dn1w, dn1b, dWqkv, dbq, dWp, dbp, dn2w, dn2b, dW1, db1, dW2, db2, dxs, dbv) = gd.out
        if ready is not None:
            db2 = ready if b2_dst is None else b2_dst.copy_(ready)
        _publish_colsum(dx, dxs)
        return (dx, dn1w, dn1b, dWqkv, dbq, dbv, dWp, dbp, dn2w, dn2b, dW1, db1, dW2, db2, None, None, None)


class HeadRegionFn(Function):
    """HeadFn as one devias_head_fwd / _bwd call"""

    @staticmethod
    def forward(ctx, slots, hw, hb, w0, b0, w2, b2, w4, b4, cdt, drop_mask=None):
        lib = _L.load()
        dev = slots.device
        slots = slots.contiguous()
        R, D = slots.shape
        C, h1, h2, G = hw.shape[0], w0.shape[0], w2.shape[0], w4.shape[0]
        dt = ops.dt_code(cdt)
        keep = [_WCACHE.get(w, cdt) for w in (hw, w0, w2, w4)] + [_f32(hb), _f32(b0), _f32(b2), _f32(b4)]
        ws = ops.workspace(lib.devias_head_workspace_bytes(R, D, C, h1, h2, G, dt), dev)
        a = _L.HeadArgs()
        a.R, a.D, a.C, a.h1, a.h2, a.G, a.dtype = R, D, C, h1, h2, G, dt
        (a.Wh, a.W0, a.W2, a.W4, a.bh, a.b0, a.b2, a.b4) = [t.data_ptr() for t in keep]
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        if drop_mask is not None:            # fc_dropout (modeling_slot.py:291,393): 0 / (1/keep) per element of the head's input
            drop_mask = ops._chk(drop_mask, "head.drop_mask", torch.float32)
            assert drop_mask.shape == (R, D)
            a.drop_mask = drop_mask.data_ptr()
        save = torch.empty((lib.devias_head_save_bytes(R, D, h1, h2, dt),), dtype=torch.uint8, device=dev)
        Z = torch.empty((R, C), dtype=cdt, device=dev)
        Mk = torch.empty((R, G), dtype=cdt, device=dev)
        _L.check(lib.devias_head_fwd(_ct.byref(a), slots.data_ptr(), Z.data_ptr(), Mk.data_ptr(), save.data_ptr(), ops._stream()), "devias_head_fwd")
        ctx.args = a
        ctx.keep = (keep, save, slots, Mk, drop_mask)
        ctx.x_version = slots._version
        ctx.params = (hw, hb, w0, b0, w2, b2, w4, b4)
        return Z, Mk

    @staticmethod
    def backward(ctx, dZ, dM):
        lib = _L.load()
        _region_state(ctx, "HeadRegionFn", ctx.keep[2] if ctx.keep is not None else None)
        a = ctx.args
        keep, save, slots, Mk, drop_mask = ctx.keep
        hw, hb, w0, b0, w2, b2, w4, b4 = ctx.params
        dev = slots.device
        dZ = dZ.contiguous() if dZ is not None else torch.zeros((a.R, a.C), dtype=slots.dtype, device=dev)
        dM = dM.contiguous() if dM is not None else torch.zeros_like(Mk)
        gd = _GradDest([(hw, tuple(hw.shape)), (hb, tuple(hb.shape)), (w0, tuple(w0.shape)), (b0, tuple(b0.shape)), (w2, tuple(w2.shape)), (b2, tuple(b2.shape)),
                        (w4, tuple(w4.shape)), (b4, tuple(b4.shape))], dev)
        g = _L.HeadGrads()
        (g.dWh, g.dbh, g.dW0, g.db0, g.dW2, g.db2, g.dW4, g.db4) = gd.ptrs()
        ws = ops.workspace(a.ws_bytes, dev)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        dslots = torch.empty_like(slots)
        _L.check(lib.devias_head_bwd(_ct.byref(a), slots.data_ptr(), Mk.data_ptr(), save.data_ptr(), dZ.data_ptr(), dM.data_ptr(), dslots.data_ptr(),
                                     _ct.byref(g), ops._stream()), "devias_head_bwd")
        ctx.keep = ctx.args = None
        return (dslots, *gd.out, None, None)


_AGG_MATS = ("to_q", "to_k", "to_v", "to_out_w", "ff0_w", "ff3_w")                       # -> devias_agg_layer_params.Wq Wk Wv Wo W1 W2
_AGG_VECS = ("to_out_b", "norm_w", "norm_b", "ctx_w", "ctx_b", "ff0_b", "ff3_b", "ffn_w", "ffn_b")   # -> bo norm_w norm_b ctx_w ctx_b b1 b2 ffn_w ffn_b
# gradient order of devias_agg_layer_grads (dWq dWk dWv dWo dbo dnorm_w dnorm_b dctx_w dctx_b dW1 db1 dW2 db2 dffn_w dffn_b) in _LAYER_KEYS names
_AGG_GRAD_KEYS = ("to_q", "to_k", "to_v", "to_out_w", "to_out_b", "norm_w", "norm_b", "ctx_w", "ctx_b", "ff0_w", "ff0_b", "ff3_w", "ff3_b", "ffn_w", "ffn_b")


class AggBlockRegionFn(Function):
    """AggBlockFoldFn as one devias_agg_block_fwd / _bwd call"""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, latents, last_w, last_b, meta, *layer_params):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = meta
        lib = _L.load()
        dev = x.device
        x = x.contiguous()
        D = x.shape[1]
        nset = 1 if tied else depth
        assert len(layer_params) == nset * len(_LAYER_KEYS) and depth <= _L.AGG_MAX_DEPTH
        LP = [dict(zip(_LAYER_KEYS, layer_params[i * 15:(i + 1) * 15])) for i in range(nset)]
        dt = ops.dt_code(cdt)
        a = _L.AggArgs()
        a.B, a.N, a.S, a.D, a.depth, a.tied, a.heads, a.dh, a.ff, a.dtype = B, N, S, D, depth, int(bool(tied)), heads, dh, LP[0]["ff0_w"].shape[0], dt
        a.eps_enc, a.eps_agg = eps_enc, eps_agg
        keep = [_f32(norm_w), _f32(norm_b), _f32(latents), _f32(last_w), _f32(last_b)]
        (a.norm_w, a.norm_b, a.latents, a.last_w, a.last_b) = [t.data_ptr() for t in keep]
        for i, P in enumerate(LP):
            mats = [_WCACHE.get(P[k], cdt) for k in _AGG_MATS]
            vecs = [_f32(P[k]) for k in _AGG_VECS]
            keep += mats + vecs
            for f, t in zip(_L.AGG_PARAM_FIELDS, mats + vecs):
                setattr(a.sets[i], f, t.data_ptr())
        save = torch.empty((lib.devias_agg_block_save_bytes(_ct.byref(a)),), dtype=torch.uint8, device=dev)
        a.save = save.data_ptr()
        ws = ops.workspace(lib.devias_agg_block_workspace_bytes(_ct.byref(a)), dev)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        slots = torch.empty((B * S, D), dtype=cdt, device=dev)
        attn_ptr = _ct.c_void_p()
        _L.check(lib.devias_agg_block_fwd(_ct.byref(a), x.data_ptr(), slots.data_ptr(), _ct.byref(attn_ptr), ops._stream()), "devias_agg_block_fwd")
        off = attn_ptr.value - save.data_ptr()
        # the last layer's slot softmax is a VIEW of the arena (no copy of 2.4 MB per step): whoever keeps `attn` alive keeps the whole arena alive,
        # and it is overwritten by nothing (an arena belongs to one forward); clone it to hold it beyond the step
        attn = save[off:off + B * heads * S * N * 4].view(torch.float32).view(B * heads, S, N)
        ctx.args = a
        ctx.meta = meta
        ctx.keep = (keep, save, x)
        ctx.x_version = x._version
        ctx.params = (norm_w, norm_b, latents, last_w, last_b, LP)
        return slots, attn

    @staticmethod
    def backward(ctx, dslots, dattn):
        B, N, S, depth, tied, heads, dh, eps_enc, eps_agg, cdt = ctx.meta
        lib = _L.load()
        _region_state(ctx, "AggBlockRegionFn", ctx.keep[2] if ctx.keep is not None else None)
        a = ctx.args
        keep, save, x = ctx.keep
        norm_w, norm_b, latents, last_w, last_b, LP = ctx.params
        dev = x.device
        D = a.D
        nset = 1 if tied else depth
        items = [(norm_w, (D,)), (norm_b, (D,)), (latents, tuple(latents.shape)), (last_w, (D,)), (last_b, (D,)), (None, (D,))]
        for P in LP:
            items += [(P[k], tuple(P[k].shape)) for k in _AGG_GRAD_KEYS]
        gd = _GradDest(items, dev)
        ptrs = gd.ptrs()
        g = _L.AggGrads()
        (g.dnorm_w, g.dnorm_b, g.dlatents, g.dlast_w, g.dlast_b, g.dx_colsum) = ptrs[:6]
        for i in range(nset):
            for f, ptr in zip(_L.AGG_GRAD_FIELDS, ptrs[6 + 15 * i:6 + 15 * (i + 1)]):
                setattr(g.sets[i], f, ptr)
        ws = ops.workspace(a.ws_bytes, dev)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        scr = _scratch(lib.devias_agg_block_scratch_bytes(_ct.byref(a)), dev)
        dslots = dslots.contiguous()
        dattn = dattn.contiguous() if dattn is not None else None
        dx = torch.empty_like(x)
        _L.check(lib.devias_agg_block_bwd(_ct.byref(a), x.data_ptr(), dslots.data_ptr(), dattn.data_ptr() if dattn is not None else None, dx.data_ptr(),
                                          _ct.byref(g), scr.data_ptr(), scr.numel(), ops._stream()), "devias_agg_block_bwd")
        ctx.keep = ctx.args = None
        out = gd.out
        _publish_colsum(dx, out[5])
        flat = []
        for i in range(nset):
            gs = dict(zip(_AGG_GRAD_KEYS, out[6 + 15 * i:6 + 15 * (i + 1)]))
            flat += [gs[k] for k in _LAYER_KEYS]
        return (dx, out[0], out[1], out[2], out[3], out[4], None, *flat)


# =====================================================================================================
# module tree (parameter containers with the reference's names)
# =====================================================================================================
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., attn_head_dim=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads if attn_head_dim is None else attn_head_dim
        all_head_dim = head_dim * num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(all_head_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if head_dim != 64:
            raise ValueError(f"devias_amd MHSA kernels are built for head_dim 64 (ViT-S/B/L), got {head_dim}")


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attn_head_dim=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, attn_head_dim=attn_head_dim)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if init_values and init_values > 0:   # gamma_1/2 exist in the reference but are never applied (:136-152)
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
        else:
            self.gamma_1, self.gamma_2 = None, None

    def run(self, x, B, N, cdt, index=0, source=None):
        a = self.attn
        if a.q_bias is None:
            raise NotImplementedError("qkv_bias=False is not used by any DEVIAS entrypoint")
        ds1 = ds2 = None
        if self.training and self.drop_path_rate > 0:
            # timm 0.4.12 drop_path (modeling_slot.py:36-47): per-sample Bernoulli(keep) mask scaled by 1/keep, drawn independently
            # for the attention and the MLP branch; applied inside the residual GEMM epilogues (row_scale)
            source = source or DropoutSource()
            ds = source.path_scale(index, B, 1.0 - self.drop_path_rate, x.device)
            ds1, ds2 = ds[0], ds[1]
        meta = (B, N, a.num_heads, self.norm1.eps, cdt)
        args = (x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias, a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, meta)
        p_drop, p_attn = float(a.proj_drop.p), float(a.attn_drop.p)          # Mlp.drop.p == proj_drop.p (Block.__init__: both `drop`)
        if self.training and (p_drop > 0 or p_attn > 0):
            # nn.Dropout inside the block (modeling_slot.py:110 attn_drop, :114 proj_drop, :66 Mlp.drop): the per-kernel sequence, not the fused region
            # (no DEVIAS recipe sets these rates; the region's GEMM epilogues carry no element mask)
            source = source or DropoutSource()
            E1 = E2 = None
            if p_drop > 0:
                M, D = x.shape
                E1 = source.element_mask("proj", index, (M, D), 1.0 - p_drop, x.device)
                E2 = source.element_mask("mlp", index, (M, D), 1.0 - p_drop, x.device)
                if ds1 is not None:                                           # drop_path(dropout(.)): one mask carries both factors
                    E1 = (E1.view(B, N, D) * ds1.view(B, 1, 1)).view(M, D).contiguous()
                    E2 = (E2.view(B, N, D) * ds2.view(B, 1, 1)).view(M, D).contiguous()
                    ds1 = ds2 = None
            adrop = (1.0 - p_attn, source.attn_seed(index)) if p_attn > 0 else None
            return EncoderBlockFn.apply(*args, ds1, ds2, (E1, E2, adrop))
        return (EncoderBlockRegionFn if _REGIONS else EncoderBlockFn).apply(*args, ds1, ds2)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        img_size = (img_size, img_size) if not isinstance(img_size, (tuple, list)) else tuple(img_size)
        patch_size = (patch_size, patch_size) if not isinstance(patch_size, (tuple, list)) else tuple(patch_size)
        self.tubelet_size = int(tubelet_size)
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * (num_frames // self.tubelet_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.proj = nn.Conv3d(in_channels=in_chans, out_channels=embed_dim,
                              kernel_size=(self.tubelet_size, patch_size[0], patch_size[1]),
                              stride=(self.tubelet_size, patch_size[0], patch_size[1]))


class MLPHead(nn.Module):
    """parameter container of the reference's MLPHead (model/modeling_slot.py:23-34): fc1 -> ReLU -> fc2; the arithmetic runs in HeadMlpFn"""

    def __init__(self, in_dim, out_dim, hidden_dim):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, out_dim)
        self.act = nn.ReLU()


class MaskPredictor(nn.Module):
    def __init__(self, dim=768, out=196):
        super().__init__()
        self.decoder = nn.Sequential(nn.Linear(dim, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, out),
                                     nn.Sigmoid())
        self.act = nn.ReLU()


class PreNorm(nn.Module):
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.):
        super().__init__()
        self.activation = nn.GELU()
        self.net = nn.Sequential(nn.Linear(dim, int(dim * mult)), self.activation, nn.Dropout(dropout),
                                 nn.Linear(int(dim * mult), dim), nn.Identity())


class SlotAttention(nn.Module):
    """agg_block/attention.py:85-141 (class `Attention` there)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = context_dim if context_dim is not None else query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))


class AggregationBlock(nn.Module):
    """agg_block/agg_block.py:8-139 with the DEVIAS settings (learned queries, pre-norm, GELU FF x4, no pos-enc, last LN)."""

    def __init__(self, *, depth=4, input_channels=768, num_latents=4, latent_dim=768, weight_tie_layers=True, ff_mult=4):
        super().__init__()
        self.num_latents, self.latent_dim, self.input_dim = num_latents, latent_dim, input_channels
        self.depth, self.weight_tie_layers = depth, weight_tie_layers
        self.heads, self.dim_head = 4, 512                                        # agg_block.py:83
        self.latents = nn.Parameter(torch.randn(num_latents, latent_dim))         # agg_block.py:62
        mk_attn = lambda: PreNorm(latent_dim, SlotAttention(latent_dim, input_channels, heads=self.heads, dim_head=self.dim_head),
                                  context_dim=input_channels)
        mk_ff = lambda: PreNorm(latent_dim, FeedForward(latent_dim, mult=ff_mult))
        self.layers = nn.ModuleList([])
        cached = None
        for _ in range(depth):
            if weight_tie_layers:
                cached = cached or (mk_attn(), mk_ff())            # cache_fn: the SAME module objects every layer
                a, f = cached
            else:
                a, f = mk_attn(), mk_ff()
            self.layers.append(nn.ModuleList([a, nn.Identity(), f, nn.Identity()]))
        self.last_layer = nn.Sequential(nn.LayerNorm(latent_dim))

    def layer_params(self) -> List[torch.Tensor]:
        out = []
        for l in range(1 if self.weight_tie_layers else self.depth):
            a, _, f, _ = self.layers[l]
            out += [a.fn.to_q.weight, a.fn.to_k.weight, a.fn.to_v.weight, a.fn.to_out[0].weight, a.fn.to_out[0].bias,
                    a.norm.weight, a.norm.bias, a.norm_context.weight, a.norm_context.bias,
                    f.fn.net[0].weight, f.fn.net[0].bias, f.fn.net[3].weight, f.fn.net[3].bias, f.norm.weight, f.norm.bias]
        return out


class VisionTransformer(nn.Module):
    """Slot-DEVIAS student (model/modeling_slot.py:219-413).  Extra kwarg: compute_dtype in {'bf16', 'fp32'}."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, fc_drop_rate=0., drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, init_scale=0.,
                 all_frames=16, tubelet_size=2, use_checkpoint=False, num_latents=4, head_type='linear',
                 slot_matching_method='hard_select', num_scene_classes=365, agg_weights_tie=False, agg_depth=4,
                 slot_matching=None, compute_dtype='bf16'):
        super().__init__()
        if slot_matching is not None:          # the reference's driver passes this misspelt kwarg (run_slot_finetuning.py:386)
            slot_matching_method = slot_matching
        if slot_matching_method not in ('hard_select', 'matching'):
            raise ValueError("incorrent slot_matching_method")
        if head_type not in ('linear', 'mlp'):
            raise ValueError(f"head_type must be 'linear' or 'mlp' (modeling_slot.py:300-313), got {head_type!r}")
        for nm, v in (("fc_drop_rate", fc_drop_rate), ("drop_rate", drop_rate), ("attn_drop_rate", attn_drop_rate)):
            if not 0.0 <= float(v) < 1.0:
                raise ValueError(f"{nm} must be in [0, 1), got {v}")
        self.dropout_source = None             # None: DropoutSource() (torch's generators); tests install one with given masks
        if use_learnable_pos_emb:
            raise NotImplementedError("learnable pos-emb is not used by DEVIAS (sinusoid table only)")
        self.num_slots = num_latents
        self.num_classes = num_classes
        self.num_scene_classes = num_scene_classes
        self.num_features = self.embed_dim = embed_dim
        self.tubelet_size = tubelet_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      num_frames=all_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.use_checkpoint = use_checkpoint   # accepted and ignored: no activation checkpointing is needed in 288 GB
        self.slot_matching_method = slot_matching_method
        self.head_type = head_type
        self.select_slots_info = [[0, 0] for _ in range(self.num_slots)]
        self.pos_embed = get_sinusoid_encoding_table(num_patches, embed_dim)     # plain attribute, not in state_dict
        self._pos_cache = {}
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc_drop_rate = float(fc_drop_rate)
        self.fc_dropout = nn.Dropout(p=fc_drop_rate) if fc_drop_rate > 0 else nn.Identity()     # parameter container as in the reference (:291); applied in HeadRegionFn
        self.agg_block = AggregationBlock(num_latents=num_latents, weight_tie_layers=agg_weights_tie, depth=agg_depth,
                                          input_channels=embed_dim, latent_dim=embed_dim)
        grid = (img_size // patch_size) if not isinstance(img_size, (tuple, list)) else (img_size[0] // patch_size)
        self.mask_predictor = MaskPredictor(embed_dim, grid * grid)
        if head_type == 'linear':                                                # modeling_slot.py:300-305
            self.head = nn.Linear(embed_dim, num_classes + num_scene_classes) if num_classes > 0 else nn.Identity()
            nn.init.trunc_normal_(self.head.weight, std=.02)
            self.apply(self._init_weights)
            self.head.weight.data.mul_(init_scale)
            self.head.bias.data.mul_(init_scale)
        else:                                                                     # 'mlp': modeling_slot.py:306-313
            self.head = MLPHead(embed_dim, num_classes + num_scene_classes, hidden_dim=512) if num_classes > 0 else nn.Identity()
            nn.init.trunc_normal_(self.head.fc1.weight, std=.02)
            nn.init.trunc_normal_(self.head.fc2.weight, std=.02)
            self.apply(self._init_weights)
            self.head.fc2.weight.data.mul_(init_scale)
            self.head.fc2.bias.data.mul_(init_scale)
        self.set_compute_dtype(compute_dtype)

    # ---- reference surface -------------------------------------------------------------------------------
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

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def get_select_slot_info(self):
        print("action slot : " + " | ".join(str(s[0]) for s in self.select_slots_info))
        print("scene slot : " + " | ".join(str(s[1]) for s in self.select_slots_info))

    def reset_select_slot_info(self):
        self.select_slots_info = [[0, 0] for _ in range(self.num_slots)]

    def set_compute_dtype(self, compute_dtype):
        table = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, torch.bfloat16: torch.bfloat16,
                 'fp32': torch.float32, 'float32': torch.float32, torch.float32: torch.float32}
        if compute_dtype not in table:
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp32', got {compute_dtype!r}")
        self.compute_dtype = table[compute_dtype]
        return self

    # ---- forward -----------------------------------------------------------------------------------------
    def _pos(self, device, dtype):
        key = (str(device), dtype)
        if key not in self._pos_cache:
            self._pos_cache[key] = self.pos_embed[0].to(device=device, dtype=dtype).contiguous()
        return self._pos_cache[key]

    def _prep_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("devias_amd.VisionTransformer runs on an MI355X only (HIP kernels; no CPU fallback): "
                               "move the model and the input to cuda")
        B, C, T, H, W = x.shape
        assert H == self.patch_embed.img_size[0] and W == self.patch_embed.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.patch_embed.img_size[0]}*{self.patch_embed.img_size[1]})."
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()          # fp16 clips (samples.half(), engine_for_slot.py:108) are widened losslessly
        return x.contiguous()

    def forward_features(self, x, return_attn=False):
        if return_attn:
            raise NotImplementedError("per-block attention maps are never materialised by the fused MHSA kernel")
        x = self._prep_input(x)
        B = x.shape[0]
        N = self.patch_embed.num_patches
        cdt = self.compute_dtype
        pe = self.patch_embed
        h = PatchEmbedFn.apply(x, pe.proj.weight, pe.proj.bias, self._pos(x.device, cdt), (pe.tubelet_size, pe.patch_size[0], cdt))
        src = self.dropout_source or DropoutSource()
        if self.training and self.pos_drop.p > 0:                                 # pos_drop (modeling_slot.py:280,356)
            h = DropMaskFn.apply(h, src.element_mask("pos", -1, tuple(h.shape), 1.0 - float(self.pos_drop.p), h.device))
        for i, blk in enumerate(self.blocks):
            h = blk.run(h, B, N, cdt, i, src)
        return h          # [B*N, D], BEFORE the final LayerNorm (it is fused into AggBlockFn)

    def forward(self, x, return_attn=False):
        B = x.shape[0]
        N = self.patch_embed.num_patches
        D = self.embed_dim
        cdt = self.compute_dtype
        h = self.forward_features(x, return_attn)
        ab = self.agg_block
        S = ab.num_latents
        meta = (B, N, S, ab.depth, ab.weight_tie_layers, ab.heads, ab.dim_head, self.norm.eps, ab.last_layer[0].eps, cdt)
        fold = _AGG_FOLD and S <= 4 and D in (384, 512, 768, 1024)
        slots, attn = ((AggBlockRegionFn if _REGIONS else AggBlockFoldFn) if fold else AggBlockFn).apply(h, self.norm.weight, self.norm.bias, ab.latents, ab.last_layer[0].weight,
                                                                      ab.last_layer[0].bias, meta, *ab.layer_params())
        if self.slot_matching_method == 'hard_select':
            raise NotImplementedError("only slot_matching_method='matching' is on the DEVIAS training path "
                                      "(the reference's hard_select branch returns empty lists, modeling_slot.py:388)")
        mp = self.mask_predictor.decoder
        drop_mask = None
        if self.training and self.fc_drop_rate > 0:
            # nn.Dropout(fc_drop_rate) on the head's input only (modeling_slot.py:393): element-wise Bernoulli(keep) / keep, drawn like drop_path's masks
            keep = 1.0 - self.fc_drop_rate
            drop_mask = ((keep + torch.rand((B * S, D), device=x.device, dtype=torch.float32)).floor() / keep).contiguous()
        if self.head_type == 'mlp':
            slots_head, mask_predictions = HeadMlpFn.apply(slots, self.head.fc1.weight, self.head.fc1.bias, self.head.fc2.weight, self.head.fc2.bias,
                                                           mp[0].weight, mp[0].bias, mp[2].weight, mp[2].bias, mp[4].weight, mp[4].bias, cdt, drop_mask)
        else:
            head_fn = HeadRegionFn if (_REGIONS or drop_mask is not None) else HeadFn
            head_args = (slots, self.head.weight, self.head.bias, mp[0].weight, mp[0].bias, mp[2].weight, mp[2].bias, mp[4].weight, mp[4].bias, cdt)
            slots_head, mask_predictions = head_fn.apply(*head_args, drop_mask) if drop_mask is not None else head_fn.apply(*head_args)
        idx = ops.slot_select(slots_head.detach(), B, S, self.num_classes).long()      # modeling_slot.py:395-401
        ar = torch.arange(B, device=x.device)
        sv, hv = slots.view(B, S, D), slots_head.view(B, S, -1)
        action_feat, scene_feat = sv[ar, idx[:, 0]], sv[ar, idx[:, 1]]
        action_logit, scene_logit = hv[ar, idx[:, 0]], hv[ar, idx[:, 1]]
        return (action_feat, scene_feat), (action_logit, scene_logit, attn), (slots_head, slots, mask_predictions)


@register_model
def slot_vit_base_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


@register_model
def slot_vit_small_patch16_224(pretrained=False, **kwargs):
    """not in the reference (its wrapper hard-wires 768); needed by BASELINE config 1"""
    model = VisionTransformer(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


@register_model
def slot_vit_large_patch16_224(pretrained=False, **kwargs):
    """not in the reference; BASELINE config 4 (D=1024, 24 blocks, 16 heads)"""
    model = VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model
