#!/usr/bin/env python3
"""Which Python lines of this package launch the ATen / runtime kernels of a training step (copies, adds, fills, memsets)?
One profiled step (torch.profiler, with_stack) of bench.py's supervised workload; prints GPU time and launch count per
(operator, innermost frame inside the repository)."""
import collections, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
from activezero_amd.nets.psmnet.psmnet_3 import PSMNet
torch.manual_seed(1)
model = PSMNet(192).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999))
il, ir, gt = bench.synth_batch(4, 540, 960, 192, dev, 1234)


def step():
    opt.zero_grad(set_to_none=True)
    loss = bench.disp_loss(model(il, ir), gt, 192)
    loss.backward()
    opt.step()


for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in e.kernels) if hasattr(e, "kernels") else 0.0
    if dt <= 0:
        continue
    # attribution: the chain of enclosing operators up to the first one that is not an aten op (a custom autograd Function of this
    # package, an optimizer step, or nothing = plain module code / the autograd engine's own accumulation)
    chain, par = [], e.cpu_parent
    while par is not None:
        chain.append(par.name)
        if not par.name.startswith("aten::"):
            break
        par = par.cpu_parent
    owner = next((c for c in chain if not c.startswith("aten::")), "(top level)")
    outer = next((c for c in reversed(chain) if c.startswith("aten::")), e.name)
    agg[(e.name, outer, owner)][0] += 1
    agg[(e.name, outer, owner)][1] += dt
tot = 0.0
for (name, outer, owner), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{us / 1e3:7.3f} ms {n:4d} x  {name:26s} via {outer:26s} in {owner[:70]}")
    tot += us
print(f"listed: {tot / 1e3:.2f} ms")
