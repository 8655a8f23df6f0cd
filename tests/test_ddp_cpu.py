"""CPU, world size 2, backend gloo: the data-parallel plumbing bench.py uses on RCCL
(activezero_amd/dist.py).  The HIP kernels cannot run here, so a small Conv3d+BN model
stands in for PSMNet; what is checked is the sharding / all-reduce / timing logic."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny():
    torch.manual_seed(1)
    return nn.Sequential(nn.Conv3d(2, 4, 3, padding=1, bias=False), nn.BatchNorm3d(4), nn.ReLU(),
                         nn.Conv3d(4, 1, 3, padding=1, bias=False))


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(2, 2, 4, 6, 6, generator=g), torch.randn(2, 1, 4, 6, 6, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from activezero_amd import dist as azdist

    r, lr, w = azdist.init("gloo")
    assert (r, lr, w) == (rank, rank, world)
    model = _tiny()
    net = azdist.wrap(model, torch.device("cpu"))
    assert isinstance(net, nn.parallel.DistributedDataParallel)
    x, y = _batch(azdist.rank_seed(1234, rank))
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    azdist.fence()
    slowest = azdist.max_over_ranks(1.0 + rank)  # rank 1 reports 2.0
    out.put((rank, [p.grad.numpy().copy() for p in model.parameters()], slowest,
             x[0, 0, 0, 0, :3].numpy().copy()))
    azdist.fence()
    azdist.shutdown()


@pytest.mark.timeout(300)
def test_ddp_two_ranks_gloo_matches_single_process_average():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((out.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, g0, t0, x0), (r1, g1, t1, x1) = results
    # every rank sees the max step time; ranks drew different pairs
    assert t0 == t1 == 2.0
    assert not (x0 == x1).all()
    # DDP averages the per-rank gradients; BatchNorm statistics stay per-rank (no SyncBN),
    # so the expectation is the mean of two independent single-rank backward passes
    expect = None
    for rank in range(world):
        m = _tiny()
        x, y = _batch(1234 + rank)
        ((m(x) - y) ** 2).mean().backward()
        gs = [p.grad for p in m.parameters()]
        expect = gs if expect is None else [a + b for a, b in zip(expect, gs)]
    expect = [g / world for g in expect]
    for a, b, e in zip(g0, g1, expect):
        assert (a == b).all()
        assert torch.allclose(torch.from_numpy(a), e, rtol=1e-5, atol=1e-7)


def test_world_size_one_is_a_no_op():
    from activezero_amd import dist as azdist

    m = _tiny()
    assert azdist.wrap(m) is m
    assert azdist.max_over_ranks(0.25) == 0.25
    azdist.fence()
