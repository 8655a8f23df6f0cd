#!/usr/bin/env python3
"""Host time to ENQUEUE one training step (forward, backward, Adam) against the time the GPU needs for it:
22 ms vs 124 ms at B=4 -- the step is not launch bound."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from activezero_amd.nets.psmnet.psmnet_3 import PSMNet
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = PSMNet(192).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=2e-4)
il, ir, gt = bench.synth_batch(4, 540, 960, 192, dev, 1234)
def step():
    opt.zero_grad(set_to_none=True)
    loss = bench.disp_loss(model(il, ir), gt, 192)
    t_f = time.perf_counter()
    loss.backward()
    t_b = time.perf_counter()
    opt.step()
    return t_f, t_b
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_f, t_b = step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host: fwd enqueue {1e3*(t_f-t0):6.1f} ms, bwd enqueue {1e3*(t_b-t_f):6.1f} ms, opt {1e3*(t1-t_b):5.1f} ms | host total {1e3*(t1-t0):6.1f} ms, GPU done at {1e3*(t2-t0):6.1f} ms")
