"""World-size-2 gloo tests (CPU) of the data-parallel plumbing: bucketed gradient all-reduce == mean of per-rank gradients
== single-process gradient of the mean of the per-rank losses (SURVEY.md §8e equivalence test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.LayerNorm(32),
                               torch.nn.Linear(32, 5))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from devias_amd.parallel import GradSync, init_distributed_from_env, broadcast_parameters
    r, _, w = init_distributed_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model = _make_model()
    if rank != 0:
        for p in model.parameters():
            p.data.add_(1.0)               # de-synchronise, then broadcast must restore rank 0's weights
    broadcast_parameters(model)
    sync = GradSync(model, bucket_bytes=512)         # several buckets
    assert len(sync.buckets) > 2
    # every parameter's place starts on a 256-byte boundary of its bucket (the [5]-element bias in front must not misalign what follows: the
    # HIP weight-gradient kernels write straight into these views and take their vector paths only for aligned destinations)
    for p in model.parameters():
        v, flat = sync._view[p], sync.flat[sync._where[p]]
        assert (v.data_ptr() - flat.data_ptr()) % 256 == 0 and flat.data_ptr() % 256 == sync._all.data_ptr() % 256
    out = []
    for step in range(2):                             # two steps: buffers are reused, p.grad stays a bucket view
        g = torch.Generator().manual_seed(100 + rank + 10 * step)
        x = torch.randn(8, 16, generator=g)
        model.zero_grad(set_to_none=True)
        model(x).pow(2).mean().backward()
        sync.finish()
        out.append([p.grad.detach().numpy().copy() for p in model.parameters()])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process comparator: gradient of the mean over ranks of the per-rank losses
    for step in range(2):
        model = _make_model()
        loss = 0
        for rank in range(world):
            g = torch.Generator().manual_seed(100 + rank + 10 * step)
            loss = loss + model(torch.randn(8, 16, generator=g)).pow(2).mean() / world
        loss.backward()
        for pi, p in enumerate(model.parameters()):
            for rank in range(world):
                assert torch.allclose(torch.from_numpy(res[rank][step][pi]), p.grad, rtol=1e-5, atol=1e-7), (step, pi, rank)


def test_gradsync_single_process_is_identity():
    from devias_amd.parallel import GradSync
    model = _make_model()
    sync = GradSync(model, bucket_bytes=4096)
    x = torch.randn(4, 16)
    model(x).sum().backward()
    sync.finish()
    ref = _make_model()
    ref(x).sum().backward()
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a.grad, b.grad)
    # bucket order is reverse registration order (head first, first layer last): SURVEY.md §3.4
    first = sync.buckets[0][0]
    assert first is list(model.parameters())[-1]


# ---- SURVEY.md §8e on the REAL loss: slot model + TrainLoss 'matching' with the rank-local teacher pad-min -------------------------------
class _OracleStudent(torch.nn.Module):
    """the CPU oracle's slot model (oracle/ref_cpu.py) wrapped as a module so that GradSync can bucket its 186-name parameter list;
    the HIP model cannot run here (no GPU), the LOSS and BUCKET path under test are the same code on both"""

    def __init__(self, cfg, seed=0):
        super().__init__()
        from devias_amd import synth
        from oracle import ref_cpu
        self.cfg = cfg
        self.names = list(ref_cpu.param_shapes(cfg))
        P = synth.fill_params(ref_cpu.param_shapes(cfg), seed=seed)
        for n in self.names:
            self.register_parameter(n.replace(".", "__"), torch.nn.Parameter(P[n].clone()))

    def P(self):
        return {n: getattr(self, n.replace(".", "__")) for n in self.names}

    def loss(self, first, B):
        from devias_amd import synth
        from oracle import ref_cpu
        cfg = self.cfg
        x = synth.video(B, cfg.all_frames, cfg.img_size, seed=1000, first=first)
        y = synth.targets(B, cfg.num_classes, seed=1000, first=first)
        tl = synth.teacher_logits(B, cfg.num_scene_classes, seed=1000, first=first)
        fg = synth.fg_masks(B, cfg.num_patches, cfg.grid * cfg.grid, seed=1000, first=first)
        out = ref_cpu.student_forward(self.P(), cfg, x)
        total, logits, ld, idx = ref_cpu.train_loss(cfg, out, tl, y, fg)
        return total.sum(), ld


def _slot_cfg():
    from oracle import ref_cpu
    return ref_cpu.SlotViTConfig(embed_dim=128, num_heads=2, depth=2, all_frames=2, img_size=64, agg_depth=2, num_latents=2)


def _slot_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from devias_amd.parallel import GradSync, init_distributed_from_env
    init_distributed_from_env(backend="gloo")
    model = _OracleStudent(_slot_cfg())
    sync = GradSync(model, bucket_bytes=256 << 10, comm_dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
    assert len(sync.buckets) >= 3
    B = 2
    if mode == "accumulate":          # update_freq = 2: rank r owns clips [4r, 4r+4) as two micro-batches of 2, loss / 2 each (engine_for_slot.py:146)
        for mb in range(2):
            sync.set_accumulate(mb == 0)
            total, _ = model.loss(first=4 * rank + 2 * mb, B=B)
            (total / 2).backward()
    else:                             # rank r owns clips [2r, 2r+2) (DistributedSampler partitioning)
        total, _ = model.loss(first=B * rank, B=B)
        total.backward()
    sync.finish()
    q.put((rank, {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}, float(total.detach())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bf16"])
def test_slot_loss_gradient_allreduce_equals_chunked_single_process(mode):
    """mean_r grad_r(batch_r) over 2 ranks == single-process gradient of mean_r loss(batch_r) evaluated chunk by chunk, each chunk with its
    OWN teacher pad-min (utils/loss/train_loss.py:103-106) -- on the slot model and the matching loss, through the flat buckets.  The
    monolithic loss of the concatenated batch is NOT the comparator (its pad value is batch-global): shown to differ.  'accumulate':
    two micro-batches per rank with set_accumulate (engine update_freq = 2); 'bf16': bf16 wire format, fp32 buckets."""
    if mode == "bf16":
        try:
            torch.zeros(2, dtype=torch.bfloat16) + 1
        except Exception:                                    # pragma: no cover
            pytest.skip("no bf16 on this host")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slot_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, t = q.get(timeout=240)
        res[r] = (g, t)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    torch.set_num_threads(4)
    model = _OracleStudent(_slot_cfg())
    nchunks = 4 if mode == "accumulate" else 2
    loss = 0
    for c in range(nchunks):
        t, _ = model.loss(first=2 * c, B=2)
        loss = loss + t / nchunks
    loss.backward()
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    tol = 2e-2 if mode == "bf16" else 1e-4      # fp32 round-off: a few gradients (slot-query LayerNorm bias) are mathematically zero and hold only noise
    for n, p in model.named_parameters():
        for rank in range(world):
            err = float((torch.from_numpy(res[rank][0][n]) - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-4 * gmax)
            assert err < tol, (mode, n, rank, err)
        assert (res[0][0][n] == res[1][0][n]).all(), n      # every rank ends with the same averaged gradient
    if mode == "plain":
        mono = _OracleStudent(_slot_cfg())
        tm, ldm = mono.loss(first=0, B=4)
        _, ld0 = mono.loss(first=0, B=2)
        _, ld1 = mono.loss(first=2, B=2)
        comp = 0.5 * (float(ld0["scene_loss"]) + float(ld1["scene_loss"]))
        assert abs(float(ldm["scene_loss"]) - comp) > 1e-6 * abs(comp)      # batch-global pad-min: the monolithic B = 4 loss is a different number


def test_gradsync_zero_fills_parameters_without_gradient():
    """a bucket with an unused parameter used to raise in finish(); now the missing gradient is zero-filled and the bucket
    is reduced like the others (every rank issues the same collectives)"""
    from devias_amd.parallel import GradSync
    model = _make_model()
    extra = torch.nn.Linear(4, 4)
    wrapper = torch.nn.ModuleList([model, extra])            # `extra` never takes part in the forward
    sync = GradSync(wrapper, bucket_bytes=256)
    x = torch.randn(4, 16)
    model(x).sum().backward()
    sync.finish()
    assert all(p.grad is not None for p in wrapper.parameters())
    assert float(extra.weight.grad.abs().max()) == 0.0 and float(extra.bias.grad.abs().max()) == 0.0
    ref = _make_model()
    ref(x).sum().backward()
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a.grad, b.grad)
    # second step after a zero-filled one, and accumulation misuse is reported
    wrapper.zero_grad(set_to_none=True)
    model(x).sum().backward()
    sync.finish()
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a.grad, b.grad)
    sync.set_accumulate(True)
    with pytest.raises(RuntimeError, match="accumulation window"):
        sync.finish()


def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from devias_amd.parallel import GradSync, init_distributed_from_env
    init_distributed_from_env(backend="nccl")
    dev = torch.device("cuda", rank)
    model = _make_model().to(dev)
    sync = GradSync(model, bucket_bytes=512)
    g = torch.Generator().manual_seed(100 + rank)
    model(torch.randn(8, 16, generator=g).to(dev)).pow(2).mean().backward()
    sync.finish()
    torch.cuda.synchronize()
    q.put((rank, [p.grad.detach().cpu().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_gradsync_two_ranks_rccl():
    """same equivalence over RCCL (side-stream all-reduce); needs >= 2 GPUs, skipped on the 1-GPU box"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    model = _make_model()
    loss = 0
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        loss = loss + model(torch.randn(8, 16, generator=g)).pow(2).mean() / world
    loss.backward()
    for pi, p in enumerate(model.parameters()):
        for rank in range(world):
            assert torch.allclose(torch.from_numpy(res[rank][pi]), p.grad, rtol=1e-4, atol=1e-6)


def _slot_hip_grads(device, first_list, dtype="fp32", sync=None):
    """HIP slot model (ViT-S width, 2 frames) on `device`: backward of the matching loss over the given 2-clip chunks (mean), through `sync`
    when given"""
    from functools import partial
    from devias_amd import synth
    from devias_amd.modeling_slot import VisionTransformer
    from devias_amd.train_loss import TrainLoss
    m = VisionTransformer(patch_size=16, embed_dim=384, depth=2, num_heads=6, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=400, all_frames=2, init_scale=1e-3, num_latents=2,
                          slot_matching_method="matching", agg_weights_tie=True, agg_depth=2, compute_dtype=dtype)
    synth.fill_module_(m, seed=0)
    m = m.to(device).train()
    crit = TrainLoss(scene_criterion="KL", num_action_classes=400, slot_matching_method="matching", scene_loss_weight=4000,
                     mask_prediction_loss_weight=1.0, mask_distill_loss_weight=1.0)
    s = sync(m) if sync is not None else None
    N = m.patch_embed.num_patches
    for first in first_list:
        x = synth.video(2, 2, 224, seed=1000, first=first).to(device)
        y = synth.targets(2, 400, seed=1000, first=first).to(device)
        tl = synth.teacher_logits(2, 365, seed=1000, first=first).to(device)
        fg = tuple(t.to(device) for t in synth.fg_masks(2, N, 196, seed=1000, first=first))
        total, _, _ = crit(m, m(x), (None, tl), y, fg_mask=fg)
        (total / len(first_list)).backward()
    if s is not None:
        s.finish()
    torch.cuda.synchronize(device)
    return {n: p.grad.detach().cpu() for n, p in m.named_parameters()}


def _slot_hip_worker(rank, world, port, q, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from devias_amd.parallel import GradSync, init_distributed_from_env
    init_distributed_from_env(backend=backend)
    dev = torch.device("cuda", rank % torch.cuda.device_count())       # gloo: both ranks may share the one GPU of the box
    torch.cuda.set_device(dev)
    g = _slot_hip_grads(dev, [2 * rank], sync=lambda m: GradSync(m, bucket_bytes=1 << 20))
    q.put((rank, {n: v.numpy() for n, v in g.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_slot_model_two_ranks_equals_chunked_single_gpu(backend):
    """SURVEY.md §8e on hardware: the HIP slot model in 2 processes (rank r <- clips [2r, 2r+2), bucketed all-reduce launched from the gradient
    hooks on the side stream, encoder weight gradients written straight into the buckets) == one process evaluating the two chunks in turn.
    fp32 mode, 1e-5.  'nccl' (RCCL over xGMI) needs >= 2 GPUs; 'gloo' runs the same code with both ranks on one GPU (collective through host
    memory), so the whole data-parallel path except the transport is exercised on a 1-GPU box."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slot_hip_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    ref = _slot_hip_grads(torch.device("cuda", 0), [0, 2])
    gmax = max(float(v.abs().max()) for v in ref.values())
    for n, v in ref.items():
        for rank in range(world):
            err = float((torch.from_numpy(res[rank][n]) - v).abs().max()) / max(float(v.abs().max()), 1e-4 * gmax)
            assert err < 1e-5, (backend, n, rank, err)


def _rccl_world1_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from devias_amd.parallel import GradSync
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    for name, cd in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        made = []

        def mk(m, cd=cd):
            made.append(GradSync(m, bucket_bytes=1 << 20, comm_dtype=cd, collective_at_world1=True))
            return made[-1]
        g = _slot_hip_grads(dev, [0], sync=mk)
        assert made[0].collective_at_world1 and len(made[0].buckets) > 1
        out[name] = {n: v.numpy() for n, v in g.items()}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_gradsync_over_a_one_rank_rccl_group():
    """The bucket path over the REAL backend on a 1-GPU box (RCCL refuses two ranks on one device): a process group of one rank, every
    bucket's all-reduce issued from the gradient hooks on the side stream through ProcessGroupNCCL (its streams, Work.wait() on the compute
    stream), fp32 and bf16 wire formats.  Sum over one rank = the gradient itself: fp32 exactly, bf16 to one rounding."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    ref = _slot_hip_grads(torch.device("cuda", 0), [0])
    for n, v in ref.items():
        assert torch.equal(torch.from_numpy(res["fp32"][n]), v), n
        b = torch.from_numpy(res["bf16"][n])
        assert torch.equal(b, v.bfloat16().float()), n


@pytest.mark.gpu
def test_slot_model_gradsync_world1_on_gpu():
    """single GPU: the two-chunk gradient accumulated through GradSync's buckets (world 1: no collective, gradients written / accumulated in
    the flat fp32 buffers) == without GradSync"""
    from devias_amd.parallel import GradSync
    dev = torch.device("cuda", 0)
    ref = _slot_hip_grads(dev, [0, 2])
    got = _slot_hip_grads(dev, [0, 2], sync=lambda m: GradSync(m, bucket_bytes=1 << 20))
    for n in ref:
        assert torch.allclose(got[n], ref[n], rtol=1e-6, atol=1e-9), n


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts `torch.distributed.run` as a
    child before any GPU call, both ranks rendezvous (gloo here), and -- this container having no MI355X -- every rank stops at the
    loud "needs an MI355X" exit AFTER init_process_group; the parent relays the child's non-zero return code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DEVIAS_DIST_BACKEND"] = "gloo"
    env["DEVIAS_BENCH_TRACE_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280)
    err = r.stderr
    assert "launching 2 ranks" in err and "torch.distributed.run" in err, err[-2000:]
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0, err[-2000:]
        import json
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2
        return
    for rank in (0, 1):
        assert f"bench.py rank {rank}/2 joined the process group" in err, err[-2000:]
    # the JSON line's `rccl` object (rank-count proof for a SCALE record), as gathered through the process group itself
    import json
    proof = json.loads([ln for ln in err.splitlines() if ln.startswith("bench.py rccl: ")][-1][len("bench.py rccl: "):])
    assert proof["world"] == 2 and proof["backend"] == "gloo" and len(proof["devices"]) == 2, proof
    assert proof["devices"][0].startswith("rank 0:") and proof["devices"][1].startswith("rank 1:"), proof
    if not torch.cuda.is_available():
        assert "needs an MI355X" in err, err[-2000:]
        assert r.returncode != 0
