"""Fused AdamW on the HIP path: torch.optim.AdamW semantics (utils/optim_factory.py:132-133 of the reference creates
`optim.AdamW(parameters, **opt_args)` over the layer-decay groups of get_parameter_groups, :49-93).

Default: the whole parameter list is updated by ONE launch (devias_adamw_multi) driven by a device table of per-tensor
descriptors; the gradient 2-norm (utils/utils.py:409-421) and clip_grad_norm_ (utils/utils.py:391) are two more launches
whose result, the clip coefficient, never leaves the device.  fp32 states, decoupled weight decay, bias correction by
step; a group's `lr_scale` is applied by the training loop exactly as in the reference (engine_for_slot.py:91-96)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, ops

_REC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4"),
                 ("bc1", "<f4"), ("bc2s", "<f4"), ("reserved", "<i8")])
assert _REC.itemsize == _lib.OPT_TENSOR_BYTES


class _Plan:
    """Static part of a multi-tensor step for one set of (parameter, state) tensors: chunk lists on the device, a ring of
    pinned host tables, the device table, the partial-sum buffer."""
    RING = 4

    def __init__(self, params, device):
        self.key = tuple(id(p) for p in params)
        sizes = [p.numel() for p in params]
        ct, ci = [], []
        for t, n in enumerate(sizes):
            k = (n + _lib.OPT_CHUNK - 1) // _lib.OPT_CHUNK
            ct.append(np.full(k, t, np.int32)); ci.append(np.arange(k, dtype=np.int32))
        self.chunk_tensor = torch.from_numpy(np.concatenate(ct) if ct else np.zeros(0, np.int32)).to(device)
        self.chunk_index = torch.from_numpy(np.concatenate(ci) if ci else np.zeros(0, np.int32)).to(device)
        self.n_chunks = self.chunk_tensor.numel()
        nbytes = max(len(params), 1) * _REC.itemsize
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.slot = 0
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.partials = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=device)
        self.norm_coef = torch.ones(2, dtype=torch.float32, device=device)

    def upload(self, rec: np.ndarray):
        s = self.slot
        self.slot = (s + 1) % self.RING
        if self.events[s] is not None:
            self.events[s].synchronize()                # the copy that last read this pinned buffer has finished
        self.host[s].numpy()[: rec.nbytes] = rec.view(np.uint8).reshape(-1)
        self.table.copy_(self.host[s], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[s] = ev


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, multi_tensor=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.multi_tensor = multi_tensor
        self._plans = {}
        self._keep = None
        self.last_grad_norm = None            # device scalar tensor after a step(max_norm=...) call

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return st

    def _gather(self):
        """[(betas, eps) -> list of (p, grad, state, lr, wd)] for every parameter that has a gradient."""
        by_hyper = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW needs contiguous fp32 parameters")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                by_hyper.setdefault((tuple(group["betas"]), group["eps"]), []).append(
                    (p, g, self._state(p), float(group["lr"]), float(group["weight_decay"])))
        return by_hyper

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, max_norm=None):
        """max_norm=None: plain step.  max_norm >= 0: also compute the global gradient 2-norm (-> self.last_grad_norm, a device
        scalar) and, when max_norm > 0, scale the gradients by min(1, max_norm/(norm+1e-6)) inside the update."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_hyper = self._gather()
        if by_hyper:
            # the kernels below write parameters through raw pointers (Tensor._version does not move): compute-dtype weight copies
            # held by the model code are stale from here on (stale bf16 weights after the first step)
            from .modeling_slot import invalidate_weight_cache
            invalidate_weight_cache()
        if not self.multi_tensor:
            if max_norm is not None:
                raise RuntimeError("FusedAdamW(multi_tensor=False) has no fused gradient-norm path")
            for (betas, eps), items in by_hyper.items():
                for p, g, st, lr, wd in items:
                    st["step"] += 1
                    ops.adamw_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], lr, betas[0], betas[1], eps, wd, st["step"], grad_scale)
            return loss
        if max_norm is not None and (len(by_hyper) > 1 or grad_scale != 1.0):
            raise RuntimeError("FusedAdamW: the fused gradient norm needs one (betas, eps) setting and grad_scale == 1")
        keep = []
        for (betas, eps), items in by_hyper.items():
            params = [it[0] for it in items]
            key = (betas, eps, tuple(id(p) for p in params))
            plan = self._plans.get(key)
            if plan is None:
                plan = self._plans[key] = _Plan(params, params[0].device)
            rec = np.zeros(len(items), _REC)
            for i, (p, g, st, lr, wd) in enumerate(items):
                st["step"] += 1
                t = st["step"]
                rec[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), lr, wd,
                          1.0 - betas[0] ** t, math.sqrt(1.0 - betas[1] ** t), 0)
                keep.append(g)
            plan.upload(rec)
            coef = None
            if max_norm is not None:
                ops.grad_sumsq_multi(plan.table, plan.chunk_tensor, plan.chunk_index, plan.partials)
                ops.clip_coef(plan.partials, plan.n_chunks, float(max_norm), plan.norm_coef)
                self.last_grad_norm = plan.norm_coef[0]
                coef = plan.norm_coef[1:2] if max_norm > 0 else None
            ops.adamw_multi(plan.table, plan.chunk_tensor, plan.chunk_index, betas[0], betas[1], eps, grad_scale, coef)
        self._keep = keep                      # converted gradients stay alive until the next step replaces them
        return loss
