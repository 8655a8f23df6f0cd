#!/usr/bin/env python3
"""Run the extractor's 2-D convolution layers (forward + weight gradient) a few times for SQ counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv2d
dev = torch.device("cuda:0")
for c, h, w, dil in ((64, 136, 240, 1), (128, 136, 240, 1), (32, 272, 480, 1)):
    x = torch.randn(8, c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(c, c, 3, 3, device=dev) * 0.05).requires_grad_(True)
    for _ in range(3):
        y = conv2d.conv_same(x, wt, dil)
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("done")
