#!/usr/bin/env python3
"""Time the V0 (32->32, [B,48,136,240]) 3-D convolution kernels alone: forward (+BN partials), input gradient,
weight gradient; HIP events; TFLOP/s against the bf16x6 roofline."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d
dev = torch.device("cuda:0")
B, D, H, W, C = 4, 48, 136, 240, 32
x = torch.randn(B, D, H, W, C, device=dev)
g = torch.randn(B, D, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
A = conv3d.DEFAULT_ARITH
pk, ci, co = conv3d._pack_forward(w, conv3d.CONV_S1, A.conv)
pd = conv3d._pack(w, C, C, 27, C * 27, True, conv3d._layout(A.conv, conv3d.CONV_S1, C))
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gf = 2.0 * 27 * C * C * B * D * H * W / 1e9
# (the first kernel timed after start-up runs ~10 % slower than the same kernel timed later -- clocks / power state --,
#  which for two rounds looked like a 0.2 ms cost of the BatchNorm-partials epilogue: warm up with 40 launches first)
for _ in range(40):
    conv3d._run_gather(g, pd, conv3d.CONV_S1, C, C, A.conv, tag="dgrad")
torch.cuda.synchronize()
for name, fn in (("fwd+stats", lambda: conv3d._run_gather(x, pk, conv3d.CONV_S1, ci, co, A.conv, stats=True)),
                 ("fwd plain", lambda: conv3d._run_gather(x, pk, conv3d.CONV_S1, ci, co, A.conv)),
                 ("dgrad", lambda: conv3d._run_gather(g, pd, conv3d.CONV_S1, C, C, A.conv, tag="dgrad")),
                 ("wgrad", lambda: conv3d._wgrad(g, x, 1, C, C, "conv", A.wgrad))):
    ms = timeit(fn)
    print(f"V0 {name:10s} {ms:7.3f} ms  {gf / ms:6.1f} TFLOP/s  ({gf / ms / 416.7:.2f} of 416.7)")
# classifier convolution (32 -> 1): VALU kernels, priced against the bytes they must move
from activezero_amd.ops import _call, _p, _stream
wc = torch.randn(1, C, 3, 3, 3, device=dev) * 0.05
go = torch.randn(B, D, H, W, device=dev)
out = torch.empty(B, D, H, W, device=dev); gx = torch.empty_like(x); gwc = torch.empty_like(wc)
mb = 4.0 * (x.numel() + out.numel()) / 1e6
for name, fn in (("c1 fwd", lambda: _call("az_conv3d_c1_fwd", _p(out), _p(x), _p(wc), None, None, None, B, D, H, W, _stream())),
                 ("c1 dgrad", lambda: _call("az_conv3d_c1_dgrad", _p(gx), _p(go), _p(wc), B, D, H, W, _stream())),
                 ("c1 wgrad", lambda: _call("az_conv3d_c1_wgrad", _p(gwc), _p(x), _p(go), None, None, B, D, H, W, _stream()))):
    ms = timeit(fn)
    print(f"V0 {name:10s} {ms:7.3f} ms  {mb / ms / 1e3:6.2f} TB/s of {mb:.0f} MB")
