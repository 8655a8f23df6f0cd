#!/usr/bin/env python3
"""Where does the end-to-end disparity error come from?  Stage-by-stage comparison on the G11 input
(256x512, maxdisp 192, eval mode) of the HIP path and of the oracle in fp32 against the oracle in fp64:
extractor features, cost logits of the three heads, disparity.  Test infrastructure (uses oracle/)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from activezero_amd.nets.psmnet import psmnet_3 as psm3  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import load_bn_buffers, load_procedural, seeded  # noqa: E402

torch.set_num_threads(16)
g = np.load(os.path.join(REPO, "tests", "golden", "g11_psmnet3_d192.npz"))
md = 192
il, ir = seeded((1, 3, 256, 512), 1101, -2.0, 2.0), seeded((1, 3, 256, 512), 1102, -2.0, 2.0)


def oracle_stages(dtype):
    m = load_bn_buffers(load_procedural(po.PSMNetOracle(md, 3), "g11."), g).to(dtype).eval()
    with torch.no_grad():
        fl, fr = m.feature_extraction(il.to(dtype)), m.feature_extraction(ir.to(dtype))
        k1, k2, k3 = m.aggregate(po.build_cost_volume(fl, fr, md // 4))
        p3 = po.soft_argmin_head(k3, md, 256, 512)
    return dict(fl=fl, fr=fr, k1=k1[:, 0], k2=k2[:, 0], k3=k3[:, 0], p3=p3)


def hip_stages(arith):
    m = load_bn_buffers(load_procedural(psm3.PSMNet(md), "g11."), g).to("cuda:0").set_arithmetic(arith).eval()
    from activezero_amd import agg3d, ops
    with torch.no_grad():
        fl, fr = m.feature_extraction.forward_pair(il.cuda(), ir.cuda())
        first = agg3d.costvol_conv_bn(fl, fr, md // 4, m.dres0[0], relu=True, arith=m.arith)
        k1, k2, k3 = m._aggregate(None, first)
        p3 = ops.softargmin(k3)
    return {k: v.cpu() for k, v in dict(fl=fl, fr=fr, k1=k1, k2=k2, k3=k3, p3=p3).items()}


r64 = oracle_stages(torch.float64)
r32 = oracle_stages(torch.float32)
rows = [("oracle fp32", r32)] + [(f"hip {a}", hip_stages(a)) for a in ("bf16x6", "fp32")]
for name, st in rows:
    print(name)
    for k in ("fl", "fr", "k1", "k2", "k3", "p3"):
        e = (st[k].double() - r64[k]).abs()
        ref = r64[k].abs()
        print(f"   {k:3s} max abs err {e.max():.3e}  mean abs err {e.mean():.3e}   (ref max {ref.max():.3e}, mean {ref.mean():.3e})")
# soft-argmin alone: the HIP head on the fp64 logits rounded to fp32
from activezero_amd import ops  # noqa: E402
with torch.no_grad():
    p = ops.softargmin(r64["k3"].float().cuda().contiguous()).cpu()
e = (p.double() - r64["p3"]).abs()
print(f"hip soft-argmin on the exact logits: max {e.max():.3e} mean {e.mean():.3e}")
p = po.soft_argmin_head(r64["k3"].float()[:, None], md, 256, 512)
e = (p.double() - r64["p3"]).abs()
print(f"torch fp32 soft-argmin on the exact logits: max {e.max():.3e} mean {e.mean():.3e}")
