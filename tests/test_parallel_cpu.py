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
