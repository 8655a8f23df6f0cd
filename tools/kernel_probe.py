#!/usr/bin/env python3
"""Launch the dominant kernels of the path alone, at bench.py's shapes (B=4, V0 =
48x136x240, 32 channels), so that rocprofv3 PMC passes can price their HBM traffic:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o fetch -- python3 tools/kernel_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -o write -- python3 tools/kernel_probe.py
    python tools/pmc_summary.py <dir> --out profiles/<name>

bn_apply (pure float4 streaming, exactly 1 read + 1 write of the tensor) is the
calibration kernel for the gfx950 FETCH_SIZE/WRITE_SIZE unit corrections
(MI355X_MICROARCH.md "HBM"): its byte counts are known a priori.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402

B, D, H, W, C = int(os.environ.get("AZ_PROBE_B", 4)), 48, 136, 240, 32
REPS = 3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(B, D, H, W, C, device=dev)
    g = torch.randn(B, D, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    for _ in range(REPS):
        # calibration: y = x*scale + shift  (reads x once, writes y once)
        _call("az_bn3d_apply", _p(y), _p(x), _p(scale), _p(shift), None, 0, x.numel() // C, C, None, _stream())
        A = conv3d.DEFAULT_ARITH  # (round 4: f16x3 forward / input gradient / weight gradient)
        conv3d._conv(x, w, conv3d.CONV_S1, A.conv, stats=True)                                  # forward + BN partials
        # BatchNorm backward of the same tensor (reduce + apply; ReLU mask recomputed from raw); round 5: dx written PRE-SPLIT
        # with its bound as amax, which is what the layer's input- and weight-gradient kernels then read
        from activezero_amd import _lib
        nv = x.numel() // C
        wsb = _lib.lib().az_bn3d_bwd_workspace(nv, C)
        ws = torch.empty(wsb // 4, device=dev)
        dgm, dbt, coef, am = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, 3, device=dev), torch.zeros(1024, device=dev)
        split = int(conv3d.PRESPLIT and A.bwd16)
        dx = torch.empty_like(x)
        _call("az_bn3d_bwd", _p(dx), None, _p(dgm), _p(dbt), _p(coef), _p(ws), wsb, _p(g), None, _p(x), _p(shift), _p(scale),
              _p(scale), _p(scale), _p(shift), 1, nv, C, _p(am), split, _stream())
        conv3d._set_amax(dx, am)
        if split:
            dx.az_split = True
        conv3d._input_grad(dx, w, conv3d.CONV_S1, C, C, conv3d.F16X3 if A.bwd16 else A.conv)
        conv3d._weight_grad(x, dx, conv3d.CONV_S1, C, C, conv3d.F16X3 if A.bwd16 else A.wgrad)
        # round 4, second half: the stride-2 / transposed pair of an hourglass (conv1 32 -> 64 stride 2, conv6 64 -> 32 transposed)
        if os.environ.get("AZ_PROBE_S2", "1") == "1":
            F = conv3d.F16X3
            w1 = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05            # conv1 weight [co][ci]
            w6 = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05            # conv6 (ConvTranspose) weight [ci][co]
            xc = torch.randn(B, D // 2, H // 2, W // 2, 64, device=dev)     # a V1 tensor
            conv3d._conv(xc, w6, conv3d.DECONV_S2, F, stats=True)           # conv6 forward (K5'')
            conv3d._input_grad(xc, w1, conv3d.CONV_S2, C, 64, F)            # conv1 input gradient (K5'')
            conv3d._weight_grad(x, xc, conv3d.CONV_S2, C, 64, F)            # conv1 weight gradient (K4w-s2)
            conv3d._weight_grad(xc, g, conv3d.DECONV_S2, 64, C, F)          # conv6 weight gradient (K4w-s2)
            conv3d._conv(x, w1, conv3d.CONV_S2, F, stats=True)              # conv1 forward (K4)
        torch.cuda.synchronize()
    print("probe done", x.numel() * 4 / 1e6, "MB per tensor")


if __name__ == "__main__":
    main()
