"""Data-parallel gradient synchronisation for the slot-ViT step: one process per GPU, RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI, bucketed all-reduce overlapped with backward on a side HIP stream.

Replaces DDP / DeepSpeed ZeRO-0 gradient all-reduce of the reference (run_slot_finetuning.py:552-563):
  * parameters are bucketed in REVERSE registration order (head / mask_predictor / agg_block first, then blocks.11 ... 0,
    patch_embed last) -- the order backward produces their gradients;
  * every parameter gets the fp32 view of its place in the flat bucket (`p._devias_grad_out`): the weight-gradient kernels of the
    encoder blocks write straight into it (modeling_slot._gout), autograd adopts the view as `.grad`, and nothing is packed;
    gradients that arrive as other tensors (small agg-block / head ones) are copied into their view by the hook;
  * when the last gradient of a bucket has arrived (post-accumulate-grad hook) an event is recorded on the compute stream and the
    all-reduce is enqueued on the side stream;
  * gradient accumulation (engine `update_freq` > 1): `set_accumulate(True)` for every micro-batch but the last -- hooks then only
    home the gradients in the buckets, the collectives start during the last micro-batch's backward;
  * finish() zero-fills the gradients of parameters that received none (unused parameters: every rank must issue the same
    collectives), makes the compute stream wait for all buckets and leaves p.grad as views of the averaged flat buffers.
xGMI is point-to-point (7 links/GPU); a few large buckets (default 64 MiB) keep each ring step bandwidth-bound.
Reduction: fp32 SUM followed by a 1/world scale (exact mean of fp32 gradients) by default; `comm_dtype=torch.bfloat16` halves the
bytes on the links (196.8 MB instead of 393.7 MB at ViT-B): gradients are rounded to bf16 once, RCCL sums them with bf16 partial
results between ring steps, the mean is widened back into the fp32 bucket.  bench.py states which one it ran.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def _os_environ_flag(name: str) -> bool:
    import os
    return os.environ.get(name, "0") not in ("", "0")


class GradSync:
    def __init__(self, module: torch.nn.Module, process_group=None, bucket_bytes: int = 64 << 20, average: bool = True,
                 comm_dtype: torch.dtype = torch.float32, simulate: bool = False, check_unused: bool = False,
                 collective_at_world1: bool = False):
        """Build AFTER module.to(device): the flat buckets are allocated on the parameters' device and every parameter gets the view of its
        place in them.  `simulate=True` (world size 1 only): every bucket still goes through the whole hook / event / side-stream path and the
        collective is replaced by a same-size device copy on the side stream (bench.py --force-gradsync: what the bookkeeping and a concurrent
        bandwidth-bound kernel cost the step, measurable on ONE GPU).  `collective_at_world1=True`: a process group of ONE rank still issues every
        bucket's all-reduce on the side stream (the whole backend path -- ProcessGroupNCCL's streams, its Work handles, the bf16 wire format -- on a
        1-GPU box; RCCL refuses two ranks on one device)."""
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("GradSync: comm_dtype must be torch.float32 or torch.bfloat16")
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.comm_dtype = comm_dtype
        self.simulate = bool(simulate) and self.world == 1
        self.collective_at_world1 = bool(collective_at_world1) and self.world == 1 and dist.is_initialized()
        self.check_unused = bool(check_unused)
        self._sim_buf = {}
        params = [p for p in module.parameters() if p.requires_grad]
        devs = {p.device for p in params}
        if len(devs) != 1:
            raise RuntimeError(f"GradSync: parameters live on {len(devs)} devices ({sorted(map(str, devs))}); move the module to ONE device first")
        self.params = list(reversed(params))
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("GradSync: master parameters must be fp32")
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat: List[torch.Tensor] = []
        self._view = {}
        self._where = {}
        # every bucket is a slice of ONE allocation (each starting on a 256-byte boundary): a bucket is still its own collective, but the mean's
        # 1/world scale (and the widening of a bf16 wire format) is ONE launch over all of them in finish()
        # every parameter's place starts on a 256-byte boundary: the weight-gradient GEMMs, LayerNorm and column-sum kernels that write straight into
        # these views take their vectorised paths only for 16-byte aligned destinations (a [765] head bias in front of a [768, 3072] matrix used to push
        # six weight gradients per step onto the generic 128 x 128 kernel: +0.35 ms; measured by diffing the kernel stats of bench.py --force-gradsync)
        _al = lambda n: (n + 63) // 64 * 64  # noqa: E731
        sizes = [sum(_al(p.numel()) for p in b) for b in self.buckets]
        starts, tot = [], 0
        for n in sizes:
            starts.append(tot)
            tot += (n + 63) // 64 * 64
        self._all = torch.zeros(tot, dtype=torch.float32, device=params[0].device)
        self._starts, self._sizes = starts, sizes
        self._all_comm = None                          # bf16 wire image of _all (comm_dtype = bf16 only)
        for bi, b in enumerate(self.buckets):
            flat = self._all[starts[bi]:starts[bi] + sizes[bi]]
            self.flat.append(flat)
            off = 0
            for p in b:
                v = flat[off:off + p.numel()].view(p.shape)
                self._view[p] = v
                self._where[p] = bi
                p._devias_grad_out = v                 # destination of the weight-gradient kernels (modeling_slot._gout)
                off += _al(p.numel())
        self._comm = [None] * len(self.buckets)        # bf16 wire buffers (comm_dtype = bf16 only)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._side = None
        self._accumulate = False
        self._saved_concurrent = None
        if (self.world > 1 or self.simulate or self.collective_at_world1) and params[0].is_cuda:
            # collectives (RCCL's kernels) will run beside backward: the persistent GEMMs pull their tiles from the dynamic queues, so that a CU the
            # collective holds or slows down takes fewer tiles instead of turning into a straggler (library option gemm_concurrent)
            from . import ops
            self._saved_concurrent = ops.get_option("gemm_concurrent")      # restored by remove() (the option used to stay set for the process)
            ops.set_option("gemm_concurrent", 1)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.reset()

    # -------------------------------------------------------------------------------------------------
    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen = set()
        self._works = []

    def set_accumulate(self, flag: bool):
        """True for every micro-batch of an accumulation window except the last one (no collective is started)."""
        self._accumulate = bool(flag)

    def _on_grad(self, p: torch.nn.Parameter):
        view = self._view[p]
        if view.device != p.device:
            raise RuntimeError("GradSync: a parameter moved to another device after the buckets were built; rebuild GradSync after model.to(device)")
        bi = self._where[p]
        if self._launched[bi] and not self._accumulate and (self.world > 1 or self.simulate or self.collective_at_world1):
            # a second backward before finish(): this gradient would be added into a bucket whose all-reduce is already in flight
            raise RuntimeError("GradSync: gradient arrived for a bucket that is already being reduced -- call finish() after every backward "
                               "(or set_accumulate(True) for all micro-batches but the last)")
        if p.grad.data_ptr() != view.data_ptr():       # produced elsewhere (or accumulated by autograd into a private tensor): home it
            view.copy_(p.grad)
            p.grad = view
        if self._accumulate or p in self._seen:
            return
        self._seen.add(p)
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi: int):
        self._launched[bi] = True
        flat = self.flat[bi]
        if self.world == 1 and not self.collective_at_world1:
            if self.simulate and flat.is_cuda:         # stand-in for the collective: the same bytes moved once on the side stream
                if self._side is None:
                    self._side = torch.cuda.Stream(device=flat.device)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(flat.device))
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    buf = self._sim_buf.get(bi)
                    if buf is None:
                        buf = self._sim_buf[bi] = torch.empty_like(flat)
                    if not _os_environ_flag("DEVIAS_SIM_NOCOPY"):     # (measurement aid: hooks + events + side stream only)
                        buf.copy_(flat, non_blocking=True)
            return
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                buf = self._wire(bi, flat)
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self._works.append((w, bi, buf))
        else:
            buf = self._wire(bi, flat)
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((w, bi, buf))

    def _wire(self, bi: int, flat: torch.Tensor) -> torch.Tensor:
        if self.comm_dtype == torch.float32:
            return flat
        if self._comm[bi] is None:
            if self._all_comm is None:
                self._all_comm = torch.zeros(self._all.numel(), dtype=torch.bfloat16, device=flat.device)
            self._comm[bi] = self._all_comm[self._starts[bi]:self._starts[bi] + self._sizes[bi]]
        if flat.is_cuda:
            from . import ops
            ops.cast(flat, torch.bfloat16, out=self._comm[bi])
        else:
            self._comm[bi].copy_(flat)
        return self._comm[bi]

    def finish(self):
        """Block the COMPUTE STREAM (not the host) until every bucket is reduced, then scale to the mean."""
        if self._accumulate:
            raise RuntimeError("GradSync.finish() inside an accumulation window: call set_accumulate(False) before the last backward")
        # Buckets whose last gradient never arrived (unused parameters) are launched here, in BUCKET-INDEX order.  If the SET of unused parameters
        # differed between ranks, one rank would issue such a bucket's collective during backward and another one here: mismatched collectives
        # (hang or cross-bucket reduction).  The reference's 'matching' path has no data-dependent branches (every parameter gets a gradient
        # every step), so this costs nothing by default; models that do have rank-dependent unused parameters set `check_unused=True`, which
        # all-reduces the "launched during backward" bitmap every step (what DDP's find_unused_parameters does) and raises on disagreement.
        if self.check_unused and self.world > 1 and dist.is_initialized():
            bitmap = torch.tensor([1.0 if l else 0.0 for l in self._launched], device=self.flat[0].device)
            tot = bitmap.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.pg)
            if not torch.equal(tot, bitmap * self.world):
                raise RuntimeError("GradSync: ranks launched different gradient buckets during backward (rank-dependent unused parameters): "
                                   "their collectives would be mismatched")
        for bi, b in enumerate(self.buckets):
            if self._launched[bi]:
                continue
            for p in b:                                 # parameters without a gradient this step: zeros, so that every rank reduces every bucket
                if p not in self._seen:
                    v = self._view[p]
                    if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                        v.zero_()
                        p.grad = v
            self._launch(bi)
        for w, bi, buf in self._works:
            w.wait()                       # for NCCL/RCCL: makes the current stream wait for the collective (no host block)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        scale = (1.0 / self.world) if (self.world > 1 and self.average) else 1.0
        wire = self.comm_dtype != torch.float32 and len(self._works) > 0
        if self._all.is_cuda:
            # every bucket was reduced this step (finish() launches the ones backward did not), so the widening of a bf16 wire format and the
            # mean's 1/world are one library launch over the whole allocation (the padding between buckets is zeros on both sides)
            from . import ops
            if wire:
                ops.cast(self._all_comm, torch.float32, out=self._all, scale=scale)
            elif scale != 1.0:
                ops.cast(self._all, torch.float32, out=self._all, scale=scale)
        else:
            for w, bi, buf in self._works:
                flat = self.flat[bi]
                if buf is not flat:                     # bf16 wire format: widen the sum back into the fp32 bucket
                    flat.copy_(buf)
                if scale != 1.0:
                    flat.mul_(scale)
        self.reset()

    def remove(self):
        """Detach from the model: hooks off, the parameters' bucket views forgotten, and the library option `gemm_concurrent` back to what it was before this
        object announced concurrent kernels (otherwise every later N = 1 step of the process stays on the dynamic-queue GEMM kernel, +0.3 ms per step and no
        four-wave kernel for ViT-L)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._saved_concurrent is not None:
            from . import ops
            ops.set_option("gemm_concurrent", self._saved_concurrent)
            self._saved_concurrent = None
        for p in self.params:
            if hasattr(p, "_devias_grad_out"):
                del p._devias_grad_out


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
    from .modeling_slot import invalidate_weight_cache
    invalidate_weight_cache()              # `.data` writes do not move Tensor._version
