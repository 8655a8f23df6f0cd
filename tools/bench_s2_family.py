#!/usr/bin/env python3
"""Stand-alone timings (HIP events, after a warm-up) of the stride-2 / transposed 3-D family of the PSMNet step at B=4 --
forward, input gradient and weight gradient of hourglass conv1 (32 -> 64, V0 -> V1) and conv6 (ConvTranspose 64 -> 32,
V1 -> V0), the 64 -> 64 layers at V1 / V2 -- in the default arithmetic (f16x3) and in bf16x6, with their share of the
respective roofline (fp32-equivalent: 2500 / 3 and 2500 / 6 TFLOP/s).  VERDICT r3 item 1's table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
B = 4
q, e, s16 = (48, 136, 240), (24, 68, 120), (12, 34, 60)


def timeit(fn, n=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


PRESPLIT = "--presplit" in sys.argv  # gradient operands as the BatchNorm-backward apply pass writes them since round 5


def presplit(dy):
    """dy (fp32, channels-last) -> the pre-split tensor az_bn3d_bwd(split_out = 1) writes for it (identity BatchNorm:
    mean 0, invstd 1, gamma 1 => dx = dy - mean(dy) - xhat mean(dy xhat)), with its bound attached"""
    from activezero_amd import _lib
    from activezero_amd.ops import _call, _p, _stream
    c = dy.shape[-1]
    nv = dy.numel() // c
    raw = torch.randn_like(dy)
    wsb = _lib.lib().az_bn3d_bwd_workspace(nv, c)
    ws, dx = torch.empty(wsb // 4, device=dev), torch.empty_like(dy)
    v = [torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev)]
    small = [torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(c, 3, device=dev)]
    am = torch.zeros(conv3d.AMAX_SLOTS, device=dev)
    _call("az_bn3d_bwd", _p(dx), None, _p(small[0]), _p(small[1]), _p(small[2]), _p(ws), wsb, _p(dy), None, _p(raw), _p(v[0]), _p(v[1]),
          _p(v[2]), None, None, 0, nv, c, _p(am), 1, _stream())
    conv3d._set_amax(dx, am)
    dx.az_split = True
    return dx


x0 = torch.randn(B, *q, 32, device=dev); w0 = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
for _ in range(30): conv3d._conv(x0, w0, conv3d.CONV_S1, conv3d.F16X3)
torch.cuda.synchronize()
print(f"{'scope':44s} {'per step':>8s} {'f16x3 ms':>9s} {'of 833':>7s} {'bf16x6 ms':>10s} {'of 417':>7s}")
tot = {3: 0.0, 1: 0.0}
def case(name, per_step, mode, cin, cout, in_dims, kind):
    """kind: fwd / dgrad / wgrad of a layer cin -> cout with index map `mode` and INPUT dims in_dims"""
    w = torch.randn(*((cin, cout) if mode == conv3d.DECONV_S2 else (cout, cin)), 3, 3, 3, device=dev) * 0.05
    x = torch.randn(B, *in_dims, cin, device=dev)
    od = conv3d._out_dims(mode, *in_dims)
    dy = torch.randn(B, *od, cout, device=dev) * 1e-4
    dy_f32 = dy
    vox = B * od[0] * od[1] * od[2]
    gf = conv3d._conv_flops(B, od[0] * od[1] * od[2], cin, cout, mode) / 1e9
    row = []
    dys = presplit(dy) if (PRESPLIT and kind != "fwd" and conv3d._presplit_ok(x, dy, mode, cin, cout, kind == "dgrad", kind == "wgrad")) else None
    if dys is not None:
        name += " [pre-split dy]"
    for prec in (conv3d.F16X3, conv3d.BF16X6):
        if dys is not None:
            dy = dys if prec == conv3d.F16X3 else dy_f32
        if kind == "fwd":
            fn = lambda: conv3d._conv(x, w, mode, prec, stats=True)
        elif kind == "dgrad":
            fn = lambda: conv3d._input_grad(dy, w, mode, cin, cout, prec)
        else:
            fn = lambda: conv3d._weight_grad(x, dy, mode, cin, cout, prec)
        with torch.no_grad():
            ms = timeit(fn)
        row.append(ms)
        tot[prec] += per_step * ms
    print(f"{name:44s} x{per_step:<7d} {row[0]:9.3f} {gf / row[0] / 833.3:7.2f} {row[1]:10.3f} {gf / row[1] / 416.7:7.2f}   ({gf:.1f} GFLOP)")
if "--only-v0" in sys.argv:
    case("V0 dgrad_m0_32_32", 6, conv3d.CONV_S1, 32, 32, q, "dgrad")
    case("V0 conv_wgrad_s1_32_32", 6, conv3d.CONV_S1, 32, 32, q, "wgrad")
    case("dgrad_m2_64_32       (conv1 dgrad)", 3, conv3d.CONV_S2, 32, 64, q, "dgrad")
    case("conv_wgrad_s2_64_32  (conv1 wgrad)", 3, conv3d.CONV_S2, 32, 64, q, "wgrad")
    case("deconv_wgrad_s2_64_32 (conv6 wgrad)", 3, conv3d.DECONV_S2, 64, 32, e, "wgrad")
    sys.exit(0)
if "--only-v0-wgrad" in sys.argv:
    case("V0 conv_wgrad_s1_32_32", 6, conv3d.CONV_S1, 32, 32, q, "wgrad")
    case("conv_wgrad_s1_64_64 @V1 (conv2 wgrad)", 3, conv3d.CONV_S1, 64, 64, e, "wgrad")
    sys.exit(0)
if "--only-s2roll" in sys.argv:  # the two uses of az_conv3d_s2roll.hip
    case("conv3d_m1_32_64      (conv1 fwd)", 3, conv3d.CONV_S2, 32, 64, q, "fwd")
    case("dgrad_m1_32_64       (conv6 dgrad)", 3, conv3d.DECONV_S2, 64, 32, e, "dgrad")
    sys.exit(0)
ONLY_S2W = "--only-s2-wgrad" in sys.argv
if ONLY_S2W:
    case("conv_wgrad_s2_64_32  (conv1 wgrad)", 3, conv3d.CONV_S2, 32, 64, q, "wgrad")
    case("deconv_wgrad_s2_64_32 (conv6 wgrad)", 3, conv3d.DECONV_S2, 64, 32, e, "wgrad")
    case("conv_wgrad_s2_64_64  (conv3 wgrad)", 3, conv3d.CONV_S2, 64, 64, e, "wgrad")
    sys.exit(0)
case("conv3d_m1_32_64      (conv1 fwd)", 3, conv3d.CONV_S2, 32, 64, q, "fwd")
case("conv3d_m2_64_32      (conv6 fwd)", 3, conv3d.DECONV_S2, 64, 32, e, "fwd")
case("dgrad_m2_64_32       (conv1 dgrad)", 3, conv3d.CONV_S2, 32, 64, q, "dgrad")
case("dgrad_m1_32_64       (conv6 dgrad)", 3, conv3d.DECONV_S2, 64, 32, e, "dgrad")
case("conv_wgrad_s2_64_32  (conv1 wgrad)", 3, conv3d.CONV_S2, 32, 64, q, "wgrad")
case("deconv_wgrad_s2_64_32 (conv6 wgrad)", 3, conv3d.DECONV_S2, 64, 32, e, "wgrad")
print("six scopes, per step, alone: f16x3 %.2f ms, bf16x6 %.2f ms" % (tot[3], tot[1]))
case("conv3d_m0_64_64 @V1  (conv2 fwd)", 3, conv3d.CONV_S1, 64, 64, e, "fwd")
case("dgrad_m0_64_64 @V1   (conv2 dgrad)", 3, conv3d.CONV_S1, 64, 64, e, "dgrad")
case("conv_wgrad_s1_64_64 @V1 (conv2 wgrad)", 3, conv3d.CONV_S1, 64, 64, e, "wgrad")
case("conv3d_m1_64_64      (conv3 fwd)", 3, conv3d.CONV_S2, 64, 64, e, "fwd")
case("conv_wgrad_s2_64_64  (conv3 wgrad)", 3, conv3d.CONV_S2, 64, 64, e, "wgrad")
case("deconv_wgrad_s2_64_64 (conv5 wgrad)", 3, conv3d.DECONV_S2, 64, 64, s16, "wgrad")
case("conv3d_m2_64_64      (conv5 fwd)", 3, conv3d.DECONV_S2, 64, 64, s16, "fwd")
case("V0 conv3d_m0_32_32   (fwd + BN partials)", 6, conv3d.CONV_S1, 32, 32, q, "fwd")
case("V0 dgrad_m0_32_32", 6, conv3d.CONV_S1, 32, 32, q, "dgrad")
case("V0 conv_wgrad_s1_32_32", 6, conv3d.CONV_S1, 32, 32, q, "wgrad")
