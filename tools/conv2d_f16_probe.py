#!/usr/bin/env python3
"""f16x3 forward of the 128 -> 128 3x3 layers (layer3 / layer4: dilation 1 and 2) alone at B = 8, 136 x 240 (77 GFLOP):
HIP events over 20 launches behind 10."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv2d
dev = torch.device("cuda:0")
for c, dil in ((128, 1), (128, 2), (64, 1)):
    x = torch.randn(8, c, 136, 240, device=dev).relu_().contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    with torch.no_grad():
        f = lambda: conv2d.conv_same(x, w, dil, f16=True)
        for _ in range(10): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gf = 2.0 * 9 * c * c * 8 * 136 * 240 / 1e9
    print(f"conv 3x3 d{dil} {c}->{c} @136x240 B=8 f16x3: {ms * 1e3:.1f} us (incl. absmax + pack), {gf / ms:.0f} TFLOP/s = {gf / ms / 833.3:.2f} of 833")
