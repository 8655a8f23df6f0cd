#!/usr/bin/env python3
"""f16x3 weight gradient of the 2-D 3x3 layers alone (B = 8 images: left + right of 4 pairs): 64 -> 64 at 136 x 240 (layer2) and
32 -> 32 at 272 x 480 (firstconv / layer1); HIP events over 20 launches behind 10.  AZ_CONV2D_WGRAD_W64 selects the 64 x 64-tile kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv2d, conv3d, _lib
dev = torch.device("cuda:0")
print("AZ_CONV2D_WGRAD_W64 =", _lib.lib().az_option(b"AZ_CONV2D_WGRAD_W64"))
for c, (h, w) in ((64, (136, 240)), (32, (272, 480))):
    xr = torch.randn(8, h, w, c, device=dev).relu_()
    gr = torch.randn(8, h, w, c, device=dev) * 1e-4
    am = (conv3d.absmax(gr), conv3d.absmax(xr))
    f = lambda: conv2d._wgrad(gr, xr, c, c, c, c, 3, 3, 1, amax=am)
    for _ in range(10): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gf = 2.0 * 9 * c * c * 8 * h * w / 1e9
    print(f"wgrad 3x3 {c}->{c} @{h}x{w} B=8: {ms * 1e3:.1f} us incl. memset + unpack, {gf / ms:.0f} TFLOP/s = {gf / ms / 833.3:.2f} of 833")
