#!/usr/bin/env python3
"""Can an HBM-bound kernel sequence and an MFMA-bound kernel share the chip?  V0 shapes ([4,48,136,240,32]):
  A = the BatchNorm backward of one layer (reduce + finalize + apply: az_bn3d_bwd), main stream
  M = one V0 weight gradient (or input gradient), second stream
timed alone and together (both launched, wall time until both are done), HIP events.  If together ~ max(A, M) the two
overlap; if together ~ A + M they only take turns.  AZ_WGRAD_R16_WGS=256 caps the weight-gradient workgroups at one per CU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d, _lib
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0")
B, D, H, W, C = 4, 48, 136, 240, 32
x = torch.randn(B, D, H, W, C, device=dev); g = torch.randn(B, D, H, W, C, device=dev)
raw = torch.randn(B, D, H, W, C, device=dev); gy = torch.randn(B, D, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
A_ = conv3d.DEFAULT_ARITH
pd = conv3d._pack(w, C, C, 27, C * 27, True, conv3d._layout(A_.conv, conv3d.CONV_S1, C))
mean, invstd, gamma = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
lib = _lib.lib()
nvox = raw.numel() // C
ws_bytes = lib.az_bn3d_bwd_workspace(nvox, C)
ws = torch.empty(ws_bytes // 4, device=dev); dx = torch.empty_like(raw)
dgamma, dbeta, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, 3, device=dev)
def bn():
    _call("az_bn3d_bwd", _p(dx), None, _p(dgamma), _p(dbeta), _p(coef), _p(ws), ws_bytes, _p(gy), None, _p(raw), _p(mean), _p(invstd),
          _p(gamma), _p(scale), _p(shift), 1, nvox, C, None, 0, _stream())
def wgrad(): conv3d._wgrad(g, x, 1, C, C, "conv", A_.wgrad)
def dgrad(): conv3d._run_gather(g, pd, conv3d.CONV_S1, C, C, A_.conv, tag="dgrad")
side = torch.cuda.Stream()
def timed(fn_main, fn_side, n=10):
    for _ in range(2):
        if fn_main: fn_main()
        if fn_side:
            with torch.cuda.stream(side): fn_side()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if fn_side:
            with torch.cuda.stream(side): fn_side()
        if fn_main: fn_main()
    torch.cuda.current_stream().wait_stream(side)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ta = timed(bn, None); print(f"BatchNorm backward alone          {ta:7.3f} ms ({4.0 * raw.numel() * 5 / ta / 1e9:.2f} TB/s of 5 tensor passes)")
for name, fn in (("weight gradient", wgrad), ("input gradient", dgrad)):
    tm = timed(None, fn)
    tt = timed(bn, fn)
    print(f"{name} alone {tm:7.3f} ms; beside the BatchNorm backward: {tt:7.3f} ms per pair (sum {ta + tm:.3f}, max {max(ta, tm):.3f}; "
          f"hidden {100 * (ta + tm - tt) / min(ta, tm):.0f} % of the shorter)")
    t3 = timed(lambda: (bn(), bn(), bn()), fn)
    print(f"    three BatchNorm backwards per {name}: {t3:7.3f} ms (sum {3 * ta + tm:.3f})")
