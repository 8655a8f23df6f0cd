#!/usr/bin/env python3
"""Launch the soft-argmin head (K6) and the patch reprojection loss (K8) alone at bench.py's shapes
(B = 4, 544x960, D = 192, ps = 11) for rocprofv3 SQ counter passes: what bounds them (VALU issue, LDS,
waiting) -- see tools/pmc_r02.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, d, h, w = 4, 48, 136, 240
logits = (3.0 * torch.randn(B, d, h, w, device=dev)).requires_grad_()
pl = (torch.rand(B, 1, 4 * h, 4 * w, device=dev) < 0.25).float()
pr = pl.roll(-17, 3).contiguous()
disp = (17.0 + 2.0 * torch.rand(B, 1, 4 * h, 4 * w, device=dev)).requires_grad_()
mask = torch.rand(B, 1, 4 * h, 4 * w, device=dev) < 0.8
for _ in range(3):
    p = ops.softargmin(logits)
    p.backward(torch.ones_like(p))
    loss, _, _ = ops.patch_reprojection(pl, pr, disp, mask, 11, want_vis=False)
    loss.backward()
    torch.cuda.synchronize()
print("aux probe done")
