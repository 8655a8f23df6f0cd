#!/usr/bin/env python3
"""three launches of the f16x3 V0 weight gradient (B = 4, 48 x 136 x 240, 32 -> 32) for the SQ counter passes of
tools/pmc_stall_passes.sh (PROBE=pmc_sq_probe_wgrad_f16.py; AZ_WGRAD_R16_WIDE selects the kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
x = torch.randn(4, 48, 136, 240, 32, device=dev)
g = torch.randn(4, 48, 136, 240, 32, device=dev) * 1e-4
with torch.no_grad():
    for _ in range(3):
        conv3d._weight_grad(x, g, conv3d.CONV_S1, 32, 32, conv3d.F16X3)
torch.cuda.synchronize()
print("done")
