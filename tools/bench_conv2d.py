#!/usr/bin/env python3
"""Per-layer timing of the 2-D convolution family at the bench shape (B = 8 images = 4 stereo pairs,
544x960): forward, input gradient and weight gradient of every distinct extractor layer geometry, HIP
events, TFLOP/s against the bf16x6 roofline (2500/6 = 416.7).  With --miopen the same layers through
torch (MIOpen fp32) for comparison (measurement tooling, not the product)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv2d  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--miopen", action="store_true")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
CL = torch.channels_last
# (name, cin, cout, k, dil, H, W, count per forward of the extractor)
LAYERS = [("firstconv/layer1 32->32", 32, 32, 3, 1, 272, 480, 8), ("layer2 64->64", 64, 64, 3, 1, 136, 240, 31),
          ("layer3.0 64->128", 64, 128, 3, 1, 136, 240, 1), ("layer3 128->128", 128, 128, 3, 1, 136, 240, 5),
          ("layer4 128->128 d2", 128, 128, 3, 2, 136, 240, 6), ("lastconv 320->128", 320, 128, 3, 1, 136, 240, 1),
          ("layer3.down 1x1 64->128", 64, 128, 1, 1, 136, 240, 1), ("lastconv.2 1x1 128->32", 128, 32, 1, 1, 136, 240, 1)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.reps


tot = [0.0, 0.0, 0.0]
print(f"{'layer':28s} {'GFLOP':>7s} | {'fwd ms':>7s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
for name, cin, cout, k, dil, h, w, cnt in LAYERS:
    x = torch.randn(8, cin, h, w, device=dev).contiguous(memory_format=CL)
    wt = (torch.randn(cout, cin, k, k, device=dev) * 0.05)
    gy = torch.randn(8, cout, h, w, device=dev).contiguous(memory_format=CL)
    gflop = 2.0 * k * k * cin * cout * 8 * h * w / 1e9
    pad = dil * (k - 1) // 2
    if args.miopen:
        torch.backends.cudnn.benchmark = True
        f = lambda: F.conv2d(x, wt, None, 1, pad, dil)
        d = lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [dil, dil], False, [0, 0], 1, [True, False, False])
        g = lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [dil, dil], False, [0, 0], 1, [False, True, False])
    else:
        xr, gr = conv2d.rows(x), conv2d.rows(gy)
        T = k * k
        pf = conv2d._pack(wt, cin, cout, cin, cout, cin * T, T, k, k, False)
        pd = conv2d._pack(wt, cout, cin, cout, cin, T, cin * T, k, k, True)
        f = lambda: conv2d._run(xr, pf, cin, cout, k, k, dil)
        d = lambda: conv2d._run(gr, pd, cout, cin, k, k, dil)
        g = lambda: conv2d._wgrad(gr, xr, cout, cin, cout, cin, k, k, dil)
    ms = [timeit(f), timeit(d), timeit(g)]
    for i in range(3):
        tot[i] += cnt * ms[i]
    print(f"{name:28s} {gflop:7.1f} | " + " | ".join(f"{m:8.3f} {gflop / m:6.1f}" for m in ms))
print(f"extractor total per step (x layer counts): fwd {tot[0]:.2f} ms, dgrad {tot[1]:.2f} ms, wgrad {tot[2]:.2f} ms, sum {sum(tot):.2f} ms")
