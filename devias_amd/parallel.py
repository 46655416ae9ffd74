"""Data-parallel gradient synchronisation for the slot-ViT step: one process per GPU, RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI, bucketed all-reduce overlapped with backward on a side HIP stream.

Replaces DDP / DeepSpeed ZeRO-0 gradient all-reduce of the reference (run_slot_finetuning.py:552-563):
  * parameters are bucketed in REVERSE registration order (head / mask_predictor / agg_block first, then blocks.11 ... 0,
    patch_embed last) -- the order backward produces their gradients;
  * when the last gradient of a bucket has been accumulated (post-accumulate-grad hook) the bucket is packed into a flat
    fp32 buffer, an event is recorded on the compute stream and the all-reduce is enqueued on the side stream;
  * finish() makes the compute stream wait for all buckets and leaves p.grad as views of the averaged flat buffers.
xGMI is point-to-point (7 links/GPU); a few large buckets (default 64 MiB) keep each ring step bandwidth-bound.
Reduction is fp32 SUM followed by a 1/world scale (exact mean of fp32 gradients), stated in DESIGN.md.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, module: torch.nn.Module, process_group=None, bucket_bytes: int = 64 << 20, average: bool = True):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = list(reversed(params))
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in self.params:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._where = {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self._where[p] = (bi, off)
                off += p.numel()
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._side = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.reset()

    # -------------------------------------------------------------------------------------------------
    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._works = []

    def _flat_for(self, bi: int, like: torch.Tensor) -> torch.Tensor:
        if self.flat[bi] is None or self.flat[bi].device != like.device:
            n = sum(p.numel() for p in self.buckets[bi])
            self.flat[bi] = torch.empty(n, dtype=torch.float32, device=like.device)
        return self.flat[bi]

    def _on_grad(self, p: torch.nn.Parameter):
        bi, off = self._where[p]
        flat = self._flat_for(bi, p.grad)
        view = flat[off:off + p.numel()].view_as(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi: int):
        flat = self.flat[bi]
        if self.world == 1:
            return
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self._works.append((w, flat))
        else:
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((w, flat))

    def finish(self):
        """Block the COMPUTE STREAM (not the host) until every bucket is reduced, then scale to the mean."""
        for bi, n in enumerate(self._pending):
            if n != 0 and n != len(self.buckets[bi]):
                raise RuntimeError(f"GradSync: bucket {bi} has {n} parameters without a gradient this step")
        for w, flat in self._works:
            w.wait()                       # for NCCL/RCCL: makes the current stream wait for the collective (no host block)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if self.world > 1 and self.average:
            for w, flat in self._works:
                flat.mul_(1.0 / self.world)
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()


def init_distributed_from_env(backend: Optional[str] = None):
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (utils/utils.py:249-282 of the reference,
    minus its hard-coded NCCL-or-exit).  Returns (rank, local_rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DEVIAS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """DDP-constructor equivalent: one broadcast of every parameter/buffer from rank 0 (SURVEY.md §2.4 last row)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
