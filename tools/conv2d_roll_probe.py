#!/usr/bin/env python3
"""az_conv2d_roll_* against torch's fp64 convolution and against az_conv2d_fwd (timing, after a warm-up)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import _lib, conv2d
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0")
lib = _lib.lib()
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def pack_roll(w, cin, cout, s_out, s_in, flip):
    pk = torch.empty(int(lib.az_conv2d_roll_packed_floats(cin, cout)), device=dev)
    _call("az_conv2d_roll_pack", _p(pk), _p(w), cin, cout, s_out, s_in, int(flip), _stream())
    return pk
warm = torch.randn(8, 136, 240, 128, device=dev)
for _ in range(200): warm.mul_(1.0)
for (B, H, W, cin, cout) in ((8, 272, 480, 32, 32), (8, 136, 240, 64, 64), (8, 136, 240, 32, 64), (2, 37, 53, 64, 32)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    res = torch.randn(B, H, W, cout, device=dev)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    pk = pack_roll(w, cin, cout, cin * 9, 9, False)
    out = torch.empty(B, H, W, cout, device=dev)
    _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), None, None, None, 0, B, H, W, cin, cout, _stream())
    e0 = float((out.double() - ref).abs().max())
    _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), None, None, _p(res), 1, B, H, W, cin, cout, _stream())
    e1 = float((out.double() - (ref + res.double()).clamp_min(0)).abs().max())
    # the library's current kernel
    pk_old = conv2d._pack(w, cin, cout, cin, cout, cin * 9, 9, 3, 3, False)
    old = conv2d._run(x, pk_old, cin, cout, 3, 3, 1)
    e_old = float((old.double() - ref).abs().max())
    e32 = float((torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).double() - ref).abs().max())
    # statistics
    groups = 2 if B % 2 == 0 else 1
    rows = int(lib.az_conv2d_roll_stats_rows(groups, B, H, W, cin, cout))
    part = torch.empty(groups, cout, rows, 2, device=dev); cnt = torch.empty(groups, rows, device=dev)
    _call("az_conv2d_roll_fwd_stats", _p(out), _p(part), _p(cnt), _p(x), _p(pk), groups, B, H, W, cin, cout, _stream())
    torch.cuda.synchronize()
    e2 = float((out.double() - ref).abs().max())
    n = cnt.double().sum(1)                                            # [G]
    mean = part[..., 0].double().sum(2) / n[:, None]                  # [G, C]
    tile_mean = part[..., 0].double() / cnt.double().clamp_min(1)[:, None, :]
    m2 = part[..., 1].double().sum(2) + (cnt.double()[:, None, :] * (tile_mean - mean[:, :, None]) ** 2).sum(2)
    rg = ref.view(groups, B // groups, H, W, cout)
    mean_ref = rg.mean(dim=(1, 2, 3)); var_ref = rg.var(dim=(1, 2, 3), unbiased=False)
    es = float(((mean - mean_ref).abs() / var_ref.sqrt()).max()); ev = float((m2 / n[:, None] / var_ref - 1).abs().max())
    gf = 2.0 * 9 * cin * cout * B * H * W / 1e9
    t_new = timeit(lambda: _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), None, None, None, 0, B, H, W, cin, cout, _stream()))
    t_st = timeit(lambda: _call("az_conv2d_roll_fwd_stats", _p(out), _p(part), _p(cnt), _p(x), _p(pk), groups, B, H, W, cin, cout, _stream()))
    t_old = timeit(lambda: conv2d._run(x, pk_old, cin, cout, 3, 3, 1))
    print(f"[{B},{H},{W}] {cin}->{cout}: max|err| plain {e0:.2e} res+relu {e1:.2e} stats-run {e2:.2e} (current kernel {e_old:.2e}, torch fp32 {e32:.2e}); "
          f"count {int(n.sum())} of {B * H * W}, mean err {es:.1e} sigma, var rel err {ev:.1e}")
    print(f"      roll {t_new * 1e3:6.1f} us = {gf / t_new:6.1f} TFLOP/s ({gf / t_new / 416.7:.2f}); +stats {t_st * 1e3:6.1f} us; current {t_old * 1e3:6.1f} us ({gf / t_old / 416.7:.2f})")
