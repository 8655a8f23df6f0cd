"""GPU, world size 2 (both ranks on cuda:0, backend gloo -- one card here, the RCCL run is the driver's):
DistributedDataParallel around a PSMNet whose weight gradients run on the side stream (overlap.py).  DDP's
reducer hooks sit behind each parameter's AccumulateGrad, which the graph construction delays until after the
join, so the all-reduced gradients must equal the plain average of the ranks' in-order gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grads(model):
    return {k: p.grad.detach().double().cpu().numpy() for k, p in model.named_parameters()}


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from activezero_amd import dist as azdist
    from activezero_amd.nets.psmnet.psmnet_3 import PSMNet
    from oracle import psmnet_oracle as po
    from tests._weights import load_procedural, seeded

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    azdist.init("gloo")
    md = 32
    il, ir = (seeded((1, 3, 256, 256), 500 + 10 * rank + i, -2.0, 2.0).to(dev) for i in range(2))
    gt = 1.0 + 28.0 * seeded((1, 1, 256, 256), 50 + rank, 0.0, 1.0).to(dev)

    def run(overlap_on, ddp):
        model = load_procedural(PSMNet(md), "g4.").to(dev).train().set_weight_grad_overlap(overlap_on)
        net = azdist.wrap(model, dev) if ddp else model
        loss = po.psmnet_disp_loss(net(il, ir), gt, po.disparity_mask(gt, md))
        loss.backward()
        torch.cuda.synchronize()
        return _grads(model)

    g_ddp = run(True, True)       # DDP + side-stream weight gradients
    g_local = run(False, False)   # this rank's own in-order gradients
    # plain average of the ranks' local gradients, by hand
    avg = {}
    for k, v in g_local.items():
        t = torch.from_numpy(v.copy())
        dist.all_reduce(t)
        avg[k] = (t / world).numpy()
    out.put((rank, {k: float(np.linalg.norm(g_ddp[k] - avg[k]) / (np.linalg.norm(avg[k]) + 1e-30)) for k in avg},
             float(sum(np.abs(v).sum() for v in g_ddp.values()))))
    azdist.fence()
    azdist.shutdown()


@pytest.mark.timeout(600)
def test_ddp_reduces_side_stream_weight_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((out.get(timeout=500) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, d0, s0), (_, d1, s1) = results
    assert abs(s0 - s1) <= 1e-9 * abs(s0)  # both ranks hold the same reduced gradients
    worst = max(max(d0.values()), max(d1.values()))
    assert worst <= 2e-4, sorted(d0.items(), key=lambda kv: -kv[1])[:3]
