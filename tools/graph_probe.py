#!/usr/bin/env python3
"""Whole-step hipGraph capture of the configs[1] training step (fwd + bwd + Adam): eager vs graph-replay step
time on the same model state and inputs, and bit-equality of the losses the two produce.
Runs the capture in THIS process; call it under `timeout` from a parent that keeps stderr."""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from activezero_amd.nets.psmnet.psmnet_3 import PSMNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1)
md, h, w = 192, 540, 960
model = PSMNet(md).to(dev).train()
state0 = copy.deepcopy(model.state_dict())
il, ir, gt = bench.synth_batch(args.batch, h, w, md, dev, 1234)


def make_opt(capturable):
    return torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), capturable=capturable)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


# ---- eager -------------------------------------------------------------------------------
opt = make_opt(False)


def eager_step():
    opt.zero_grad(set_to_none=True)
    loss = bench.disp_loss(model(il, ir), gt, md)
    loss.backward()
    opt.step()
    return loss.detach().clone()


for _ in range(2):
    eager_step()
ms_eager, _ = timed(eager_step, args.steps)
print(f"eager        {ms_eager:8.2f} ms/step", flush=True)

# reference losses from the initial state, eager
model.load_state_dict(state0)
opt = make_opt(False)
ref = [eager_step().item() for _ in range(4)]

# ---- captured ----------------------------------------------------------------------------
model.load_state_dict(state0)
opt = make_opt(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
scratch_state = copy.deepcopy(model.state_dict())
with torch.cuda.stream(side):  # warm-up on a side stream (allocator + optimizer state), then restore the state
    for _ in range(2):
        eager_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
model.load_state_dict(scratch_state)
for st in opt.state.values():  # Adam moments / step back to zero
    for k, v in st.items():
        if torch.is_tensor(v):
            v.zero_()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
print("capturing", flush=True)
with torch.cuda.graph(graph):
    static_loss = bench.disp_loss(model(il, ir), gt, md)
    static_loss.backward()
    opt.step()
print("captured", flush=True)


def graph_step():
    graph.replay()
    return static_loss


got = []
for _ in range(4):
    graph.replay()
    got.append(static_loss.item())
print("losses eager:", ref)
print("losses graph:", got)
print("bit-equal:", ref == got, flush=True)
ms_graph, _ = timed(graph_step, args.steps)
print(f"graph replay {ms_graph:8.2f} ms/step  ({ms_eager - ms_graph:+.2f} ms vs eager)", flush=True)
