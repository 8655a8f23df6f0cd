#!/usr/bin/env python3
"""A few launches of az_conv3d_s2roll.hip at the B = 4 hourglass size (conv1 forward with BatchNorm partials, conv6 input
gradient) for the SQ counter passes of tools/pmc_stall_passes.sh (PROBE=pmc_sq_probe_s2roll.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
x = torch.randn(4, 48, 136, 240, 32, device=dev)
w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
for _ in range(4):
    conv3d._conv(x, w, conv3d.CONV_S2, conv3d.F16X3, stats=True)
    conv3d._input_grad(x, w, conv3d.DECONV_S2, 64, 32, conv3d.F16X3)
torch.cuda.synchronize()
