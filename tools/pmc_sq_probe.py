#!/usr/bin/env python3
"""Run the V0 32->32 forward conv a few times (precision from AZ_CONV_PRECISION) for SQ counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
x = torch.randn(4, 48, 136, 240, 32, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk, ci, co = conv3d._pack_forward(w, 0, conv3d.DEFAULT_ARITH.conv)
for _ in range(3):
    conv3d._run_gather(x, pk, 0, ci, co, conv3d.DEFAULT_ARITH.conv, stats=True)
torch.cuda.synchronize()
print("done")
