#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
x = torch.randn(4, 48, 136, 240, 32, device=dev)
g = torch.randn(4, 48, 136, 240, 32, device=dev)
for _ in range(3):
    conv3d._wgrad(g, x, 1, 32, 32, "conv", conv3d.DEFAULT_ARITH.wgrad)
torch.cuda.synchronize()
print("done")
