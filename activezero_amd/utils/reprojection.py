"""Warp / reprojection operators -- drop-in for the reference module
utils/reprojection.py (same function names, arguments and return tuples).

  apply_disparity ................ az_warp_gather_{fwd,bwd}   (K7)
  get_reproj_error_patch ......... az_patch_reproj_{fwd,bwd}  (K8, fused)
  local_contrast_norm ............ az_lcn                     (K9)
  get_reprojection_error[_old|_diff_ratio]: compositions of K1/K7 as in the
  reference (utils/reprojection.py:38-96, 130-173); their masked means are
  computed without boolean-index compaction.
"""
import torch
import torch.nn.functional as F

from activezero_amd import ops
from .warp_ops import apply_disparity_cu


def apply_disparity(img, disp):
    """Bilinear sample of img at (x + disp) with the reference's grid convention
    (utils/reprojection.py:13-35); differentiable w.r.t. disp and img."""
    return ops.warp_gather(img, disp)


def _masked_mse(a, b, mask):
    """F.mse_loss(a[mask], b[mask]) without the compaction / host sync."""
    m = mask.to(a.dtype)
    return ((a - b) ** 2 * m).sum() / m.sum()


def get_reprojection_error(input_L, input_R, pred_disp_l, pred_disp_r, mask_l=None, mask_r=None):
    input_L_warped = apply_disparity(input_R, -pred_disp_l)
    input_R_warped = apply_disparity(input_L, pred_disp_r)
    if mask_l is None:
        disp_gt_l = apply_disparity_cu(pred_disp_r.detach().contiguous(),
                                       pred_disp_r.detach().type(torch.int).contiguous())
        disp_gt_r = apply_disparity_cu(pred_disp_l.detach().contiguous(),
                                       (-pred_disp_l.detach().type(torch.int)).contiguous())
        mask_l = ((disp_gt_l < 192) * (disp_gt_l > 0)).detach()
        mask_r = ((disp_gt_r < 192) * (disp_gt_r > 0)).detach()
    c = input_L.shape[1]
    mask_l = mask_l.repeat(1, c, 1, 1)
    mask_r = mask_r.repeat(1, c, 1, 1)
    return (_masked_mse(input_L_warped, input_L, mask_l),
            _masked_mse(input_R_warped, input_R, mask_r),
            input_L_warped, input_R_warped, mask_l.type(torch.int), mask_r.type(torch.int))


def get_reprojection_error_old(input_L, input_R, pred_disp_l, mask=None):
    input_L_warped = apply_disparity(input_R, -pred_disp_l)
    if mask is not None:
        mask = mask.repeat(1, input_L.shape[1], 1, 1)
    else:
        mask = torch.ones_like(input_L_warped).type(torch.bool)
    return _masked_mse(input_L_warped, input_L, mask), input_L_warped, mask.type(torch.int)


def get_reproj_error_patch(input_L, input_R, pred_disp_l, mask=None, ps=5):
    assert ps % 2 == 1
    return ops.patch_reprojection(input_L, input_R, pred_disp_l, mask, ps)


def get_reprojection_error_diff_ratio(input_L, input_R, pred_disp_l, mask=None):
    ratio = [0.25, 0.5, 1]
    weight = [0.3, 0.5, 0.2]
    if mask is not None:
        mask = mask.repeat(1, input_L.shape[1], 1, 1)
    else:
        mask = torch.ones_like(input_L)
    mask = mask.type(torch.float32).detach()
    output, loss_dict, total_loss = {}, {}, 0
    for i, (r, wgt) in enumerate(zip(ratio, weight)):
        resize = lambda t: F.interpolate(t, scale_factor=r, mode="bilinear")
        tgt, src = resize(input_L).contiguous(), resize(input_R).contiguous()
        disp_rs = resize(pred_disp_l) * r
        mask_rs = resize(mask).type(torch.bool)
        warped = apply_disparity(src, (-disp_rs).contiguous())
        loss = _masked_mse(warped, tgt, mask_rs)
        output[f"stage{i}"] = {"target": tgt, "warped": warped, "pred_disp": disp_rs,
                               "mask": mask_rs.type(torch.int)}
        loss_dict[f"stage{i}"] = loss.item()
        total_loss = total_loss + loss * wgt
    return total_loss, output, loss_dict


def local_contrast_norm(image, kernel_size=9, eps=1e-5):
    """(x - mean_k) / (std_k + eps) over a zero-padded k x k window; returns
    (normed, std) like utils/reprojection.py:175-200."""
    assert kernel_size % 2 == 1, "Kernel size should be odd"
    if image.shape[1] > 1:
        image = image[:, :1, :, :]
    assert image.shape[1] == 1, "Only support single channel image for now"
    return ops.local_contrast_norm(image.contiguous(), kernel_size, eps)
