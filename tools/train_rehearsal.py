#!/usr/bin/env python3
"""train.py:220-432 (train_sample + train_sample_onreal) restated for the keys this path consumes, driven by
the synthetic MessytableDataset-shaped loader: DataLoader -> sim step (psmnet_disp + patch reprojection) ->
real step (patch reprojection), ADAPTER=False (3-channel PSMNet; the adapter is outside the path).  The
reference's own train.py imports yacs / tensorboardX / torchvision, none of which exist in this image, so
this script is the executable rehearsal of "drops into train.py": the same calls in the same order on the
same item dictionary.

    python tools/train_rehearsal.py --iters 4 --batch 2
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd.datasets.messytable_synthetic import SyntheticMessytableDataset  # noqa: E402
from activezero_amd.nets.psmnet.psmnet_3 import PSMNet  # noqa: E402
from activezero_amd.utils import disp_losses  # noqa: E402
from activezero_amd.utils.reprojection import get_reproj_error_patch  # noqa: E402
from activezero_amd.utils.warp_ops import apply_disparity_cu  # noqa: E402

MAX_DISP, PATCH = 192, 11  # configs/config.py:12, 41


def train_sample(sample, model, opt):
    """train.py:237-312 (sim) and :369-419 (real)"""
    model.train()
    img_L, img_R = sample["img_sim_L"], sample["img_sim_R"]
    half = lambda t: F.interpolate(t, scale_factor=0.5, mode="nearest", recompute_scale_factor=False)
    img_disp_r = half(sample["img_disp_R"])                                   # :261-265
    disp_gt_l = apply_disparity_cu(img_disp_r, img_disp_r.type(torch.int))    # :266-268
    mask = (disp_gt_l < MAX_DISP) * (disp_gt_l > 0)                           # :272
    opt.zero_grad()
    out = model(img_L, img_R)
    sim_loss = disp_losses.psmnet_disp(out, disp_gt_l, mask)                  # losses.py:162-183
    reproj, _, _ = get_reproj_error_patch(sample["img_sim_L_reproj"], sample["img_sim_R_reproj"], out[0], mask, PATCH)
    sim_loss = sim_loss + reproj                                              # losses.py:95-97
    sim_loss.backward()
    opt.step()
    opt.zero_grad()
    out = model(sample["img_real_L"], sample["img_real_R"])
    real_loss, _, _ = get_reproj_error_patch(sample["img_real_L_reproj"], sample["img_real_R_reproj"], out[0], None, PATCH)
    real_loss.backward()
    opt.step()
    return float(sim_loss.detach()), float(real_loss.detach())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ds = SyntheticMessytableDataset(length=a.iters * a.batch, device=dev)
    loader = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=False, num_workers=0)
    torch.manual_seed(1)
    model = PSMNet(MAX_DISP).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for i, sample in enumerate(loader):
        s, r = train_sample(sample, model, opt)
        print(f"iter {i}: sim loss {s:.4f}  real reprojection loss {r:.5f}", flush=True)


if __name__ == "__main__":
    main()
