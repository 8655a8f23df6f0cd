#!/usr/bin/env python3
"""A few launches of the V0 input gradient in both arithmetics, for the counter passes of tools/pmc_stall_passes.sh
(PROBE=pmc_f16_probe.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
B, D, H, W, C = 4, 48, 136, 240, 32
g = torch.randn(B, D, H, W, C, device=dev) * 1e-6
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
pd = conv3d._pack(w, C, C, 27, C * 27, True, conv3d._layout(conv3d.BF16X6, conv3d.CONV_S1, C))
for _ in range(6):
    conv3d._run_gather(g, pd, conv3d.CONV_S1, C, C, conv3d.BF16X6, tag="dgrad")
    conv3d._input_grad_f16(g, w, conv3d.CONV_S1, C, C)
torch.cuda.synchronize()
