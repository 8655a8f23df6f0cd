"""Disparity loss -- replacement for `psmnet_disp` of the reference's utils/losses.py:7-15, computed
by the fused K12 kernel (az_disp_loss_{fwd,bwd}): one pass over the maps, no boolean-index
compaction, no host sync.

This module is deliberately NOT named utils/losses.py: the reference's module of that name also
holds the `AllLosses` driver class that train.py imports (control plane, depends on the yacs
config) and must keep resolving from the reference tree.  Integration is one line there:
`from utils.disp_losses import psmnet_disp`.
"""
from activezero_amd import ops


def psmnet_disp(pred_disp, disp_gt_l, mask):
    """0.5*SL1(pred1) + 0.7*SL1(pred2) + SL1(pred3), means over `mask` (bool, shape of disp_gt_l).
    pred_disp = (pred3, pred2, pred1) as PSMNet.forward returns them in training mode."""
    pred3, pred2, pred1 = pred_disp
    return ops.disp_loss(pred3, pred2, pred1, disp_gt_l, mask)


def psmnet_disp_range(pred_disp, disp_gt_l, max_disp, min_disp=0.0):
    """Same loss with the mask rule of train.py:272 (min_disp < gt < max_disp) applied in the kernel,
    so the caller builds no mask tensor at all."""
    pred3, pred2, pred1 = pred_disp
    return ops.disp_loss(pred3, pred2, pred1, disp_gt_l, None, float(min_disp), float(max_disp))
