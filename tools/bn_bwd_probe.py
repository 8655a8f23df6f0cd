#!/usr/bin/env python3
"""az_bn3d_bwd alone on the V0 tensor ([4,48,136,240,32]) and on a 64-channel one; run under rocprofv3 --kernel-trace --stats
for the per-kernel split (reduce / finalize / apply)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import _lib
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0")
lib = _lib.lib()
def run(shape, relu, n=20):
    C = shape[-1]
    raw = torch.randn(*shape, device=dev); gy = torch.randn(*shape, device=dev); dx = torch.empty_like(raw)
    mean, invstd, gamma = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    nvox = raw.numel() // C
    ws_bytes = lib.az_bn3d_bwd_workspace(nvox, C)
    ws = torch.empty(ws_bytes // 4, device=dev)
    dgamma, dbeta, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, 3, device=dev)
    def bn():
        _call("az_bn3d_bwd", _p(dx), None, _p(dgamma), _p(dbeta), _p(coef), _p(ws), ws_bytes, _p(gy), None, _p(raw), _p(mean), _p(invstd),
              _p(gamma), _p(scale) if relu else None, _p(shift) if relu else None, int(relu), nvox, C, None, 0, _stream())
    for _ in range(3): bn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): bn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    gb = 4.0 * raw.numel() * 5 / 1e9
    print(f"bn3d_bwd {shape} relu={relu}: {ms:.3f} ms, {gb / ms:.2f} TB/s over 5 tensor passes ({gb:.2f} GB)")
run((4, 48, 136, 240, 32), True)
run((4, 48, 136, 240, 32), False)
run((4, 24, 68, 120, 64), True)
run((4, 24, 68, 120, 64), False)
