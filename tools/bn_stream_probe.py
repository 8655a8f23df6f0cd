#!/usr/bin/env python3
"""HBM rate of the BatchNorm apply stream on a V0 tensor (802 MB in, 802 MB out) next to torch's copy kernel on
the same box (the yardstick for what a read+write stream reaches here)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from activezero_amd import ops
dev = torch.device("cuda:0")
C = 32
xv = torch.randn(4, 48, 136, 240, C, device=dev); yv = torch.empty_like(xv)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ms = timeit(lambda: ops._call("az_bn3d_apply", yv.data_ptr(), xv.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, xv.numel() // C, C, None, ops._stream()))
print(f"bn_apply V0: {ms:.3f} ms  {8.0 * xv.numel() / ms / 1e9:.2f} TB/s")
ms = timeit(lambda: yv.copy_(xv))
print(f"torch copy : {ms:.3f} ms  {8.0 * xv.numel() / ms / 1e9:.2f} TB/s")
