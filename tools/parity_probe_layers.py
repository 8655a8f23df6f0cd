#!/usr/bin/env python3
"""Per-layer error of the extractor's conv+BN units (eval mode): every unit is fed the EXACT (fp64 oracle)
input rounded to fp32; its output is compared with the fp64 output -- HIP kernel vs torch fp32 on the CPU."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from activezero_amd import conv2d  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_3 as psm3  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import load_bn_buffers, load_procedural, seeded  # noqa: E402

torch.set_num_threads(16)
g = np.load(os.path.join(REPO, "tests", "golden", "g11_psmnet3_d192.npz"))
md = 192
il = seeded((1, 3, 256, 512), 1101, -2.0, 2.0)
m64 = load_bn_buffers(load_procedural(po.PSMNetOracle(md, 3), "g11."), g).double().eval()
m32 = load_bn_buffers(load_procedural(po.PSMNetOracle(md, 3), "g11."), g).eval()
hip = load_bn_buffers(load_procedural(psm3.PSMNet(md), "g11."), g).to("cuda:0").eval()
rec = {}
names = {}
for n, mod in m64.feature_extraction.named_modules():
    if isinstance(mod, torch.nn.Sequential) and len(mod) == 2 and isinstance(mod[1], torch.nn.BatchNorm2d):
        names[mod] = n
        mod.register_forward_hook(lambda mod, inp, out: rec.__setitem__(names[mod], (inp[0].detach().clone(), out.detach().clone())))
with torch.no_grad():
    m64.feature_extraction(il.double())
mods32 = dict(m32.feature_extraction.named_modules())
modsh = dict(hip.feature_extraction.named_modules())
print(f"{'unit':28s} {'hip max':>10s} {'hip mean':>10s} {'t32 max':>10s} {'t32 mean':>10s}   |y| mean")
with torch.no_grad():
    for n, (x, y) in rec.items():
        u32, uh = mods32[n], modsh[n]
        y32 = u32(x.float())
        xh = x.float().cuda().contiguous(memory_format=torch.channels_last)
        yh = conv2d.conv_bn_eval(xh, uh[0], uh[1])
        if yh is None:
            from activezero_amd import bn2d
            yh = bn2d.bn_act(conv2d.conv(xh, uh[0]), uh[1])
        eh, e32 = (yh.cpu().double() - y).abs(), (y32.double() - y).abs()
        print(f"{n:28s} {eh.max():10.2e} {eh.mean():10.2e} {e32.max():10.2e} {e32.mean():10.2e}   {y.abs().mean():.2e}")

# SPP upsampling written as two GEMMs vs F.interpolate (fp64)
from activezero_amd.nets.psmnet import psmnet_submodule_3 as sub  # noqa: E402
import torch.nn.functional as F  # noqa: E402
with torch.no_grad():
    for hw in ((1, 2), (2, 4), (4, 8), (8, 16)):
        t = torch.randn(2, 32, *hw, dtype=torch.float64)
        ref = F.interpolate(t, (64, 128), mode="bilinear", align_corners=True)
        got = sub.upsample_bilinear_ac(t.float().cuda(), (64, 128)).cpu().double()
        t32 = F.interpolate(t.float(), (64, 128), mode="bilinear", align_corners=True).double()
        print(f"upsample {hw}: hip(matmul) max {float((got - ref).abs().max()):.2e}   torch fp32 max {float((t32 - ref).abs().max()):.2e}")
