#!/usr/bin/env python3
"""Round-4 probe: three matrix instructions per product instead of six.

f16x3 = operands scaled by a power of two from the tensor's largest magnitude, split into two fp16 parts, products
hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, fp32 accumulation (az_roll_common.h).  Measured here on the real
kernels, per layer, against the fp64 result: this library's bf16x6 kernel, the f16x3 kernel and torch's own fp32
convolution (CPU), on operands with the statistics of a backward pass (heavy-tailed gradients of magnitude ~1e-6,
weights ~0.05).  Then the timing of the V0 input gradient in both arithmetics.

    python tools/f16x3_probe.py [--no-time]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(4)
torch.set_num_threads(16)


def err(got, ref):
    e = (got.double().cpu() - ref).abs()
    scale = ref.abs().mean()
    return float(e.mean() / scale), float(e.max() / scale), float((got.double().cpu() - ref).mean() / scale)


def dgrad_ref(dy, w, dtype):
    # input gradient of a stride-1 conv = conv with flipped, channel-swapped kernel
    wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
    return F.conv3d(dy.permute(0, 4, 1, 2, 3).to(dtype), wt.to(dtype), padding=1).permute(0, 2, 3, 4, 1).contiguous()


ACC = "--time-only" not in sys.argv
print("== per-layer error of the input gradient against fp64 (mean / max of |err| over mean |ref|; signed mean) ==")
print(f"{'case':46s} {'bf16x6':>28s} {'f16x3':>28s} {'torch fp32 (CPU)':>28s}")
for name, cout, cin, gscale, tail, wscale in (
        ("K=864   dy~1e-6 heavy-tailed, w~0.05", 32, 32, 1e-6, 2.0, 0.05),
        ("K=864   dy~1 gaussian, w~0.05", 32, 32, 1.0, 0.0, 0.05),
        ("K=864   dy~1e-3, 1e4 dynamic range, w~1e-3", 32, 32, 1e-3, 3.0, 1e-3),
        ("K=1728  dy~1e-6 heavy-tailed, w~0.05 (64 ch)", 64, 32, 1e-6, 2.0, 0.05),
        ("K=864   dy~1e-30 (tiny), w~0.05", 32, 32, 1e-30, 1.0, 0.05)) if ACC else ():
    # layer: cin -> cout forward; its input gradient maps dy [.., cout] -> dx [.., cin]
    dy = torch.randn(1, 12, 40, 48, cout) * gscale * torch.exp(tail * torch.randn(1, 12, 40, 48, cout))
    w = torch.randn(cout, cin, 3, 3, 3) * wscale
    ref = dgrad_ref(dy.double(), w.double(), torch.float64)
    t32 = dgrad_ref(dy, w, torch.float32)
    dyg, wg = dy.to(dev), w.to(dev)
    pk = conv3d._pack(wg, cout, cin, 27, cin * 27, True, conv3d._layout(conv3d.BF16X6, conv3d.CONV_S1, cin))
    x6 = conv3d._run_gather(dyg, pk, conv3d.CONV_S1, cout, cin, conv3d.BF16X6, tag="dgrad")
    h3 = conv3d._input_grad_f16(dyg, wg, conv3d.CONV_S1, cin, cout)
    dyg.az_amax = None  # (the next case is a new tensor anyway)
    cells = ["%.2e / %.2e (%+.1e)" % err(v, ref) for v in (x6, h3, t32)]
    print(f"{name:46s} {cells[0]:>28s} {cells[1]:>28s} {cells[2]:>28s}")

if "--no-time" in sys.argv:
    sys.exit(0)

print("== V0 input gradient, B=4 [4,48,136,240,32], HIP events over 10 launches after 40 warm-up launches ==")
B, D, H, W, C = 4, 48, 136, 240, 32
g = torch.randn(B, D, H, W, C, device=dev) * 1e-6
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
pd = conv3d._pack(w, C, C, 27, C * 27, True, conv3d._layout(conv3d.BF16X6, conv3d.CONV_S1, C))
pk16, wam = conv3d._pack_f16(w, C, C, 27, C * 27, True, conv3d.CONV_S1)
gam = conv3d.absmax(g)
out = torch.empty_like(g)
from activezero_amd.ops import _call, _p, _stream  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


f6 = lambda: conv3d._run_gather(g, pd, conv3d.CONV_S1, C, C, conv3d.BF16X6, tag="dgrad")
f3 = lambda: _call("az_conv3d_fwd_f16", _p(out), _p(g), _p(pk16), _p(gam), _p(wam), 0, None, None, None, 0, 0, B, C, C, D, H, W, _stream())
fa = lambda: _call("az_absmax", _p(gam), _p(g), g.numel(), _stream())
for _ in range(40):
    f6()
torch.cuda.synchronize()
gf = 2.0 * 27 * C * C * B * D * H * W / 1e9
for name, fn in (("bf16x6", f6), ("f16x3", f3), ("bf16x6 again", f6), ("f16x3 again", f3)):
    ms = timeit(fn)
    print(f"V0 dgrad {name:14s} {ms:7.3f} ms  {gf / ms:6.1f} TFLOP/s fp32-equivalent")
ms = timeit(fa)
print(f"az_absmax of the same tensor: {ms:.3f} ms = {4e-9 * g.numel() / ms:.2f} TB/s")
