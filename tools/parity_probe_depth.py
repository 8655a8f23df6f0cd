#!/usr/bin/env python3
"""Error growth along the extractor's depth (eval mode, real propagated inputs): output of every residual
block, HIP vs the fp32 oracle, both against the fp64 oracle."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from activezero_amd.nets.psmnet import psmnet_3 as psm3  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import load_bn_buffers, load_procedural, seeded  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(16)
g = np.load(os.path.join(REPO, "tests", "golden", "g11_psmnet3_d192.npz"))
il = seeded((1, 3, 256, 512), 1101, -2.0, 2.0)


def blocks(fe):
    return {n: m for n, m in fe.named_modules() if n.count(".") == 1 and n.startswith("layer")}


def run(fe, x):
    rec = {}
    hs = [m.register_forward_hook(lambda m, i, o, n=n: rec.__setitem__(n, o.detach().clone().cpu().double()))
          for n, m in blocks(fe).items()]
    with torch.no_grad():
        rec["out"] = fe(x).detach().cpu().double()
    for h in hs:
        h.remove()
    return rec


m64 = load_bn_buffers(load_procedural(po.PSMNetOracle(192, 3), "g11."), g).double().eval()
m32 = load_bn_buffers(load_procedural(po.PSMNetOracle(192, 3), "g11."), g).eval()
hip = load_bn_buffers(load_procedural(psm3.PSMNet(192), "g11."), g).to("cuda:0").eval()
r64 = run(m64.feature_extraction, il.double())
r32 = run(m32.feature_extraction, il)
rh = run(hip.feature_extraction, il.cuda())
print(f"{'block':12s} {'hip mean':>10s} {'hip max':>10s} {'t32 mean':>10s} {'t32 max':>10s}  |y| mean   signed mean err: hip, t32;  64x64-pooled |err|: hip, t32")
for n in r64:
    dh, d3 = rh[n] - r64[n], r32[n] - r64[n]
    eh, e3 = dh.abs(), d3.abs()
    ph = F.avg_pool2d(dh, 64, 64).abs().mean() if dh.shape[-1] >= 64 and dh.shape[-2] >= 64 else float("nan")
    p3 = F.avg_pool2d(d3, 64, 64).abs().mean() if dh.shape[-1] >= 64 and dh.shape[-2] >= 64 else float("nan")
    print(f"{n:12s} {eh.mean():10.2e} {eh.max():10.2e} {e3.mean():10.2e} {e3.max():10.2e}  {r64[n].abs().mean():.2e}   "
          f"{dh.mean():+.2e} {d3.mean():+.2e}   {ph:.2e} {p3:.2e}")

# ---- the SPP / lastconv section, piece by piece, each fed the EXACT (fp64) input rounded to fp32 ----------
import torch.nn.functional as F  # noqa: E402
from activezero_amd import conv2d  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_submodule_3 as sub  # noqa: E402

fe64, feh = m64.feature_extraction, hip.feature_extraction
raw64, skip64 = r64["layer2.15"], r64["layer4.2"]
CL = torch.channels_last
with torch.no_grad():
    size = skip64.shape[-2:]
    skh = skip64.float().cuda().contiguous(memory_format=CL)
    pooled, p = {}, skh
    for _, win in sorted(sub._SPP_WINDOWS, key=lambda iw: iw[1]):
        p = F.avg_pool2d(p, win, win) if not pooled else F.avg_pool2d(p, 2, 2)
        pooled[win] = p
    pyr64, pyrh = [], []
    for i, win in ((4, 8), (3, 16), (2, 32), (1, 64)):
        br64 = getattr(fe64, f"branch{i}")
        pool64 = F.avg_pool2d(skip64, win, win)
        e = (pooled[win].cpu().double() - pool64).abs()
        print(f"pool {win:2d}: hip hierarchical mean {e.mean():.2e} max {e.max():.2e}   (|y| {pool64.abs().mean():.2e})")
        u64 = F.relu(br64[1](pool64))
        uh = sub._convbn_unit(pool64.float().cuda().contiguous(memory_format=CL), getattr(feh, f"branch{i}")[1], relu=True)
        e = (uh.cpu().double() - u64).abs()
        print(f"branch{i} unit on exact pool: mean {e.mean():.2e} max {e.max():.2e}")
        up64 = F.interpolate(u64, size, mode="bilinear", align_corners=True)
        uph = sub.upsample_bilinear_ac(u64.float().cuda().contiguous(memory_format=CL), size)
        e = (uph.cpu().double() - up64).abs()
        print(f"branch{i} upsample on exact unit: mean {e.mean():.2e} max {e.max():.2e}   shape {tuple(uph.shape)} strides {uph.stride()}")
        pyr64.append(up64)
    cat64 = torch.cat([raw64, skip64] + pyr64, 1)
    l0_64 = F.relu(fe64.lastconv[0](cat64))
    l0h = sub._convbn_unit(cat64.float().cuda().contiguous(memory_format=CL), feh.lastconv[0], relu=True)
    e = (l0h.cpu().double() - l0_64).abs()
    print(f"lastconv.0 on exact cat: mean {e.mean():.2e} max {e.max():.2e}")
    o64 = fe64.lastconv[2](l0_64)
    oh = conv2d.conv(l0_64.float().cuda().contiguous(memory_format=CL), feh.lastconv[2])
    o32 = m32.feature_extraction.lastconv[2](l0_64.float())
    e, e3 = (oh.cpu().double() - o64).abs(), (o32.double() - o64).abs()
    print(f"lastconv.2 on exact input: hip mean {e.mean():.2e} max {e.max():.2e}   torch fp32 mean {e3.mean():.2e} max {e3.max():.2e}")
