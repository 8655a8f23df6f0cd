"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's error metrics
(utils/cascade_metrics.py:16-62 compute_err_metric, :65-126 compute_obj_err) in plain torch
fp32, the eager op sequence the reference runs.  Pinned by tests/golden/g10_metrics.npz, which
tools/make_goldens.py produced by calling the imported reference functions.
(The loss of utils/losses.py:7-15 is restated in psmnet_oracle.psmnet_disp_loss.)
"""
import numpy as np
import torch
import torch.nn.functional as F


@torch.no_grad()
def compute_err_metric(disp_gt, depth_gt, disp_pred, focal_length, baseline, mask, depth_pred=None):
    """cascade_metrics.py:29-61."""
    epe = F.l1_loss(disp_pred[mask], disp_gt[mask], reduction="mean").item()
    dd = torch.abs(disp_gt[mask] - disp_pred[mask])
    n = dd.numel()
    bad1 = (dd > 1).sum().item() / n
    bad2 = (dd > 2).sum().item() / n
    if depth_pred is None:
        depth_pred = focal_length * baseline / disp_pred  # metres (:36-37)
    mm = torch.clip(torch.abs(depth_gt[mask] * 1000 - depth_pred[mask] * 1000), min=0, max=100)
    dz = torch.abs(depth_gt[mask] - depth_pred[mask])
    return {
        "epe": epe, "bad1": bad1, "bad2": bad2, "depth_abs_err": torch.mean(mm).item(),
        "depth_err2": (dz > 2e-3).sum().item() / n, "depth_err4": (dz > 4e-3).sum().item() / n,
        "depth_err8": (dz > 8e-3).sum().item() / n,
    }


@torch.no_grad()
def compute_obj_err(disp_gt, depth_gt, disp_pred, focal_length, baseline, label, mask, obj_total_num=17):
    """cascade_metrics.py:80-126."""
    depth_pred = focal_length * baseline / disp_pred
    out = [np.zeros(obj_total_num) for _ in range(4)]
    for obj in label.unique():
        oid = int(obj.item())
        m = (label == oid) * mask
        out[0][oid] += F.l1_loss(disp_gt[m], disp_pred[m], reduction="mean").item()
        out[1][oid] += torch.mean(torch.clip(torch.abs(depth_gt[m] * 1000 - depth_pred[m] * 1000), min=0, max=100)).item()
        dz = torch.abs(depth_gt[m] - depth_pred[m])
        out[2][oid] += (dz > 4e-3).sum().item() / dz.numel()
        out[3][oid] += 1
    return tuple(out)
