"""Fused AdamW on the HIP path (devias_adamw_step): torch.optim.AdamW semantics (utils/optim_factory.py:132-133 of the
reference creates `optim.AdamW(parameters, **opt_args)` over the layer-decay groups of get_parameter_groups, :49-93).
One launch per parameter tensor, fp32 states, decoupled weight decay, bias correction by step; `lr_scale` of a group is
applied by the training loop exactly as in the reference (engine_for_slot.py:91-96)."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous parameters")
                ops.adamw_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                               st["step"], grad_scale)
        return loss
