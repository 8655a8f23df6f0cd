"""ORACLE (test infrastructure, never the product path).

CPU restatement in eager PyTorch of the reference's warp / reprojection ops.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  Pinned by tests/golden/g6_*.npz, g7_*.npz, g8_*.npz generated from
the imported reference (tools/make_goldens.py).

Reference lines restated (relative to /root/reference):
  * apply_disparity ............... utils/reprojection.py:13-35
  * get_reprojection_error_old .... utils/reprojection.py:81-96
  * get_reproj_error_patch ........ utils/reprojection.py:99-127
  * get_reprojection_error_diff_ratio utils/reprojection.py:130-173
  * local_contrast_norm ........... utils/reprojection.py:175-200
  * get_reprojection_error ........ utils/reprojection.py:38-78 (uses the scatter
    warp of utils/warp_ops.py -> oracle/warp_oracle.py)
"""
import torch
import torch.nn.functional as F

from .warp_oracle import apply_disparity_cu_oracle


def apply_disparity(img, disp):
    """Bilinear gather of `img` at x + disp (pixels), zero padding.

    The reference builds a [0,1] linspace grid and feeds 2*g-1 to grid_sample
    with align_corners=False, which samples pixel coordinate
        px = j*W/(W-1) + d - 0.5 ,  py = i*H/(H-1) - 0.5
    (see sample_coords() below)."""
    b, _, h, w = img.shape
    gx = torch.linspace(0, 1, w, dtype=img.dtype).view(1, 1, w).expand(b, h, w)
    gy = torch.linspace(0, 1, h, dtype=img.dtype).view(1, h, 1).expand(b, h, w)
    gx = gx + (disp / w)[:, 0]
    grid = torch.stack((gx, gy), dim=3)
    return F.grid_sample(img, 2 * grid - 1, mode="bilinear", padding_mode="zeros",
                         align_corners=False)


def sample_coords(h, w, disp):
    """Closed form of the pixel coordinates apply_disparity() samples (float64)."""
    j = torch.arange(w, dtype=torch.float64).view(1, 1, w)
    i = torch.arange(h, dtype=torch.float64).view(1, h, 1)
    px = j * w / (w - 1) + disp[:, 0].double() - 0.5
    py = (i * h / (h - 1) - 0.5).expand_as(px)
    return px, py


def _masked_mse(a, b, mask):
    return F.mse_loss(a[mask], b[mask])


def get_reprojection_error_old(input_l, input_r, pred_disp_l, mask=None):
    warped = apply_disparity(input_r, -pred_disp_l)
    if mask is None:
        mask = torch.ones_like(warped, dtype=torch.bool)
    else:
        mask = mask.repeat(1, input_l.shape[1], 1, 1)
    return _masked_mse(warped, input_l, mask), warped, mask.int()


def get_reproj_error_patch(input_l, input_r, pred_disp_l, mask=None, ps=5):
    """ps x ps patches are unfolded into channels, every tap is warped with the
    CENTRE pixel's disparity, masked MSE; `warped` visualisation = Fold (sum of
    overlapping patches, not normalised), cropped back to HxW."""
    assert ps % 2 == 1
    b, c, h, w = input_l.shape
    r = (ps - 1) // 2
    taps_l = F.unfold(input_l, ps, padding=r).reshape(b, c * ps * ps, h, w)
    taps_r = F.unfold(input_r, ps, padding=r).reshape(b, c * ps * ps, h, w)
    warped = apply_disparity(taps_r, -pred_disp_l)
    if mask is None:
        mask = torch.ones_like(warped, dtype=torch.bool)
    else:
        mask = mask.repeat(1, c * ps * ps, 1, 1)
    loss = _masked_mse(warped, taps_l, mask)
    vis = F.fold(warped.reshape(b, c * ps * ps, h * w), (h + ps - 1, w + ps - 1), ps)
    if ps > 1:
        vis = vis[:, :, r:-r, r:-r]
    return loss, vis, mask[:, :c].int()


def get_reprojection_error_diff_ratio(input_l, input_r, pred_disp_l, mask=None):
    if mask is None:
        mask = torch.ones_like(input_l)
    else:
        mask = mask.repeat(1, input_l.shape[1], 1, 1)
    mask = mask.float().detach()
    stages, parts, total = {}, {}, 0
    for i, (ratio, weight) in enumerate(zip((0.25, 0.5, 1), (0.3, 0.5, 0.2))):
        rs = lambda t: F.interpolate(t, scale_factor=ratio, mode="bilinear")
        tgt, src = rs(input_l), rs(input_r)
        disp = rs(pred_disp_l) * ratio
        m = rs(mask).bool()
        warped = apply_disparity(src, -disp)
        loss = _masked_mse(warped, tgt, m)
        stages[f"stage{i}"] = {"target": tgt, "warped": warped, "pred_disp": disp,
                               "mask": m.int()}
        parts[f"stage{i}"] = loss.item()
        total = total + loss * weight
    return total, stages, parts


def get_reprojection_error(input_l, input_r, pred_disp_l, pred_disp_r, mask_l=None, mask_r=None):
    warped_l = apply_disparity(input_r, -pred_disp_l)
    warped_r = apply_disparity(input_l, pred_disp_r)
    if mask_l is None:
        gt_l = apply_disparity_cu_oracle(pred_disp_r, pred_disp_r.int())
        gt_r = apply_disparity_cu_oracle(pred_disp_l, -pred_disp_l.int())
        mask_l = ((gt_l < 192) & (gt_l > 0)).detach()
        mask_r = ((gt_r < 192) & (gt_r > 0)).detach()
    c = input_l.shape[1]
    mask_l = mask_l.repeat(1, c, 1, 1)
    mask_r = mask_r.repeat(1, c, 1, 1)
    return (_masked_mse(warped_l, input_l, mask_l), _masked_mse(warped_r, input_r, mask_r),
            warped_l, warped_r, mask_l.int(), mask_r.int())


def local_contrast_norm(image, kernel_size=9, eps=1e-5):
    assert kernel_size % 2 == 1
    image = image[:, :1]
    b, _, h, w = image.shape
    win = F.unfold(image, kernel_size, padding=(kernel_size - 1) // 2)
    mean = win.mean(dim=1).view(b, 1, h, w)
    std = win.std(dim=1, unbiased=False).view(b, 1, h, w)
    return (image - mean) / (std + eps), std
