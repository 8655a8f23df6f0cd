"""Error metrics -- drop-in for the reference's utils/cascade_metrics.py:16-62
(`compute_err_metric`): same arguments, same dict of Python floats, computed in one pass by the
K12 kernel (az_disp_metrics) and fetched with a single device->host copy instead of ten masked
reductions with an .item() each.  `compute_obj_err` (cascade_metrics.py:65-126, imported by
test.py) runs the same kernel once per object label.
"""
import numpy as np
import torch

from activezero_amd import ops


@torch.no_grad()
def compute_err_metric(disp_gt, depth_gt, disp_pred, focal_length, baseline, mask, depth_pred=None):
    """disp_gt, depth_gt, disp_pred [bs,1,H,W]; focal_length, baseline broadcastable per batch element
    (the dataset yields [bs,1,1,1], datasets/messytable.py:286-297); mask bool [bs,1,H,W]."""
    fb = None
    if depth_pred is None:
        fb = (focal_length * baseline).to(torch.float32)
        bs = disp_gt.shape[0]
        if fb.numel() == 1:
            fb = fb.reshape(1).expand(bs)
        if fb.numel() != bs:  # a full map: materialise depth_pred as the reference does
            depth_pred, fb = fb / disp_pred, None
    acc = ops.disp_metrics(disp_gt, depth_gt, disp_pred, mask, fb, depth_pred).tolist()
    n = acc[7]
    div = (lambda v: v / n) if n > 0 else (lambda v: float("nan"))
    return {
        "epe": div(acc[0]), "bad1": div(acc[1]), "bad2": div(acc[2]),
        "depth_abs_err": div(acc[3]), "depth_err2": div(acc[4]), "depth_err4": div(acc[5]),
        "depth_err8": div(acc[6]),
    }


@torch.no_grad()
def compute_obj_err(disp_gt, depth_gt, disp_pred, focal_length, baseline, label, mask, obj_total_num=17):
    """Per-object disparity / depth errors (cascade_metrics.py:65-126): returns
    (total_obj_disp_err, total_obj_depth_err, total_obj_depth_4_err, total_obj_count), numpy [obj_total_num]."""
    fb = (focal_length * baseline).to(torch.float32)
    depth_pred = (fb / disp_pred).contiguous()
    disp_err = np.zeros(obj_total_num)
    depth_err = np.zeros(obj_total_num)
    depth_4_err = np.zeros(obj_total_num)
    count = np.zeros(obj_total_num)
    accs, ids = [], []
    for obj in label.unique().tolist():
        ids.append(int(obj))
        accs.append(ops.disp_metrics(disp_gt, depth_gt, disp_pred, (label == obj) & mask.bool(), None, depth_pred))
    if accs:
        for obj_id, acc in zip(ids, torch.stack(accs).tolist()):  # one device->host copy for all objects
            n = acc[7]
            disp_err[obj_id] += acc[0] / n if n else float("nan")
            depth_err[obj_id] += acc[3] / n if n else float("nan")
            depth_4_err[obj_id] += acc[5] / n if n else float("nan")
            count[obj_id] += 1
    return disp_err, depth_err, depth_4_err, count
