#!/usr/bin/env python3
"""Which parameter gradients differ between the two-stream and the in-order backward of one full-size step
(race hunting: differences beyond float-atomic noise point at the operand a side-stream kernel lost)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd.nets.psmnet import psmnet_3 as psm3
from activezero_amd.utils.disp_losses import psmnet_disp
from tests._weights import load_procedural, seeded
DEV = "cuda:0"
il, ir = (torch.nn.functional.pad(seeded((1, 3, 540, 960), 1103 + i, -2.0, 2.0), (0, 0, 4, 0)).to(DEV) for i in range(2))
gt = seeded((1, 1, 544, 960), 1301, -12.0, 215.0).to(DEV)
mask = (gt < 192) * (gt > 0)
def grads(overlap):
    m = load_procedural(psm3.PSMNet(192), "g11.").to(DEV).train().set_weight_grad_overlap(overlap)
    psmnet_disp(m(il, ir), gt, mask).backward()
    torch.cuda.synchronize()
    return {k: p.grad.double().cpu() for k, p in m.named_parameters()}
ref = grads(False)
for trial in range(3):
    got = grads(trial > 0)  # trial 0: the in-order pass AGAIN -- the run-to-run noise of the float atomics alone
    label = "two-stream" if trial > 0 else "in-order, second run"
    bad = []
    for k in ref:
        e = float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-30))
        if e > 2e-5: bad.append((e, k))
    print(f"{label}: {len(bad)} of {len(ref)} gradients differ by more than 2e-5 from the (first) in-order pass")
    for e, k in sorted(bad, reverse=True)[:25]: print(f"   {e:.2e}  {k}")
