#!/usr/bin/env python3
"""Stand-alone timings (HIP events, after a warm-up) of every 3-D weight-gradient shape of the PSMNet step at B=4,
with their share of the bf16x6 roofline: which of them is worth rebuilding next."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
A = conv3d.DEFAULT_ARITH
B = 4
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
# warm-up: clocks
x0 = torch.randn(B, 48, 136, 240, 32, device=dev); g0 = torch.randn_like(x0)
for _ in range(20): conv3d._wgrad(g0, x0, 1, 32, 32, "conv", A.wgrad)
torch.cuda.synchronize()
rows = []
def case(name, per_step, coarse_shape, fine_shape, stride, cm, cn):
    c = torch.randn(*coarse_shape, device=dev); f = torch.randn(*fine_shape, device=dev)
    ms = timeit(lambda: conv3d._wgrad(c, f, stride, cm, cn, "conv", A.wgrad))
    gf = 2.0 * 27 * cm * cn * c.numel() / cm / 1e9
    rows.append((name, per_step, ms, gf))
    print(f"{name:44s} x{per_step}/step  {ms:7.3f} ms  {gf:7.1f} GFLOP  {gf / ms:6.1f} TFLOP/s  {gf / ms / 416.7:.2f} of 416.7   -> {per_step * ms:5.2f} ms/step")
q, e, s16 = (48, 136, 240), (24, 68, 120), (12, 34, 60)
case("V0 32->32 s1 @1/4 (wgrad16)", 6, (B, *q, 32), (B, *q, 32), 1, 32, 32)
case("hourglass conv1 32->64 s2 (coarse 1/8, fine 1/4)", 3, (B, *e, 64), (B, *q, 32), 2, 64, 32)
case("hourglass conv2 64->64 s1 @1/8", 3, (B, *e, 64), (B, *e, 64), 1, 64, 64)
case("hourglass conv3 64->64 s2 (1/16 <- 1/8)", 3, (B, *s16, 64), (B, *e, 64), 2, 64, 64)
case("hourglass conv4 64->64 s1 @1/16", 3, (B, *s16, 64), (B, *s16, 64), 1, 64, 64)
case("hourglass conv5 deconv 64->64 (1/16 -> 1/8)", 3, (B, *s16, 64), (B, *e, 64), 2, 64, 64)
case("hourglass conv6 deconv 64->32 (1/8 -> 1/4)", 3, (B, *e, 64), (B, *q, 32), 2, 64, 32)
print("total per step: %.2f ms stand-alone" % sum(p * ms for _, p, ms, _ in rows))
