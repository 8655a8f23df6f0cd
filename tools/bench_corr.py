#!/usr/bin/env python3
"""K10 alone: the RAFT-Stereo correlation volume and its two gradient GEMMs at BASELINE configs[4] ([4,256,136,240]);
HIP events after a warm-up.  The kernels are priced against the bytes they must move (the volume is 125 MB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0")
B, C, H, W = 4, 256, 136, 240
f1 = torch.randn(B, C, H, W, device=dev); f2 = torch.randn(B, C, H, W, device=dev)
corr = torch.empty(B, H, W, W, device=dev); g = torch.randn(B, H, W, W, device=dev)
g1 = torch.empty_like(f1); g2 = torch.empty_like(f2)
def timeit(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
vol = lambda: _call("az_corr1d_volume", _p(corr), _p(f1), _p(f2), B, C, H, W, W, _stream())
bw1 = lambda: _call("az_corr1d_volume_bwd", _p(g1), None, _p(g), _p(f1), _p(f2), B, C, H, W, W, _stream())
bw2 = lambda: _call("az_corr1d_volume_bwd", None, _p(g2), _p(g), _p(f1), _p(f2), B, C, H, W, W, _stream())
gf = 2.0 * B * H * W * W * C / 1e9
mb_v = 4.0 * (f1.numel() + f2.numel() + corr.numel()) / 1e6
mb_b = 4.0 * (g.numel() + f2.numel() + g1.numel()) / 1e6
for name, fn, mb in (("volume", vol, mb_v), ("d/d fmap1", bw1, mb_b), ("d/d fmap2", bw2, mb_b)):
    ms = timeit(fn)
    print(f"K10 {name:10s} {ms:7.3f} ms  {gf / ms:7.1f} TFLOP/s ({gf:.1f} GFLOP)  {mb / ms / 1e3:5.2f} TB/s of {mb:.0f} MB algorithmic "
          f"= {mb / ms / 1e3 / 8.0:.2f} of 8 TB/s")
ref = torch.einsum("bchw,bchv->bhwv", f1.double(), f2.double()) / C ** 0.5
vol(); torch.cuda.synchronize()
print("volume max |err| vs fp64:", float((corr.double() - ref).abs().max()), " (fp32 einsum:", float((torch.einsum("bchw,bchv->bhwv", f1, f2) / C ** 0.5 - ref).abs().max()), ")")
