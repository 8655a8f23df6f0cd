#!/usr/bin/env python3
"""az_bn2d_bwd / az_bn3d_bwd alone on the tensors of the step that are NOT V0-sized (the 2-D extractor's, the 1/8-resolution
64-channel volume): ms per call and TB/s over the five tensor passes.  AZ_BN_BWD_FUSED=0 for the three-kernel sequence."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import _lib
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0")
lib = _lib.lib()


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def run2d(groups, nvox, C, relu):
    raw = torch.randn(groups * nvox, C, device=dev); gy = torch.randn_like(raw); dx = torch.empty_like(raw)
    mean, invstd = torch.zeros(groups, C, device=dev), torch.ones(groups, C, device=dev)
    scale, shift = torch.ones(groups, C, device=dev), torch.zeros(groups, C, device=dev)
    gamma = torch.ones(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws_bytes = lib.az_bn2d_workspace(groups, nvox, C)
    ws = torch.empty(ws_bytes // 4, device=dev)
    am = torch.zeros(1024, device=dev)
    def f():
        _call("az_bn2d_bwd", _p(dx), None, _p(dg), _p(db), _p(ws), ws_bytes, _p(gy), None, _p(raw), _p(mean), _p(invstd), _p(gamma),
              _p(scale) if relu else None, _p(shift) if relu else None, int(relu), groups, nvox, C, _p(am), _stream())
    ms = timeit(f)
    gb = 4.0 * raw.numel() * 5 / 1e9
    print(f"bn2d_bwd groups={groups} nvox={nvox} C={C} relu={relu}: {ms * 1e3:7.1f} us  {gb / ms:5.2f} TB/s over 5 passes ({gb * 1e3:.0f} MB)")


def run3d(shape, relu):
    C = shape[-1]
    raw = torch.randn(*shape, device=dev); gy = torch.randn_like(raw); dx = torch.empty_like(raw)
    mean, invstd, gamma = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    nvox = raw.numel() // C
    ws_bytes = lib.az_bn3d_bwd_workspace(nvox, C)
    ws = torch.empty(ws_bytes // 4, device=dev)
    dg, db, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, 3, device=dev)
    am = torch.zeros(1024, device=dev)
    def f():
        _call("az_bn3d_bwd", _p(dx), None, _p(dg), _p(db), _p(coef), _p(ws), ws_bytes, _p(gy), None, _p(raw), _p(mean), _p(invstd),
              _p(gamma), _p(scale) if relu else None, _p(shift) if relu else None, int(relu), nvox, C, _p(am), 0, _stream())
    ms = timeit(f)
    gb = 4.0 * raw.numel() * 5 / 1e9
    print(f"bn3d_bwd {shape} relu={relu}: {ms * 1e3:7.1f} us  {gb / ms:5.2f} TB/s over 5 passes ({gb * 1e3:.0f} MB)")


print("AZ_BN_BWD_FUSED =", os.environ.get("AZ_BN_BWD_FUSED", "1 (default)"))
run2d(2, 4 * 136 * 240, 64, True)
run2d(2, 4 * 136 * 240, 64, False)
run2d(2, 4 * 136 * 240, 128, True)
run2d(2, 4 * 272 * 480, 32, True)
run2d(2, 4 * 136 * 240, 32, True)
run3d((4, 24, 68, 120, 64), True)
run3d((4, 12, 34, 60, 64), True)
run3d((4, 48, 136, 240, 32), True)
