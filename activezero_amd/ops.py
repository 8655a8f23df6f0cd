"""Torch-facing wrappers over the C ABI (include/azhip.h).

PyTorch is plumbing only here: it owns device memory and the current HIP
stream; every computation is a call into libazhip.so through raw pointers.
All wrappers validate like the reference does (contiguity / dtype / device /
sign, utils/warp_ops.py:69-77) and raise on violation -- there is no eager or
CPU fallback.
"""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    return t


def _call(name, *args):
    fn = getattr(_lib.lib(), name)
    _lib.check(fn(*args), name)


def _p(t):
    return t.data_ptr() if t is not None else None


# ----------------------------------------------------------------------------
# K1/K2 scatter warp
# ----------------------------------------------------------------------------
def warp_scatter(img, disp, sign):
    """dst[n,c,y,j+disp] = src[n,c,y,j] with the reference's collision rule."""
    _chk(img, "img")
    _chk(disp, "disp", torch.int32)
    n, c, h, w = img.shape
    if disp.numel() != n * h * w:
        raise RuntimeError("disp must be [N,H,W] or [N,1,H,W]")
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _call("az_warp_scatter", _p(out), _p(img), _p(disp), n, c, h, w, int(sign), _stream())
    return out


# ----------------------------------------------------------------------------
# K3 cost volume
# ----------------------------------------------------------------------------
class _CostVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_l, feat_r, ndisp, channels_last):
        feat_l = _chk(feat_l.contiguous(), "feat_l")
        feat_r = _chk(feat_r.contiguous(), "feat_r")
        if feat_l.shape != feat_r.shape:
            raise RuntimeError("feature maps must have identical shapes")
        ctx.ndisp, ctx.cl = ndisp, channels_last
        with torch.cuda.device(feat_l.device):
            if channels_last:
                b, h, w, c = feat_l.shape
                out = feat_l.new_empty(b, ndisp, h, w, 2 * c)
                _call("az_cost_volume_fwd_ndhwc", _p(out), _p(feat_l), _p(feat_r), b, c, ndisp, h, w, _stream())
            else:
                b, c, h, w = feat_l.shape
                out = feat_l.new_empty(b, 2 * c, ndisp, h, w)
                _call("az_cost_volume_fwd", _p(out), _p(feat_l), _p(feat_r), b, c, ndisp, h, w, _stream())
        ctx.dims = (b, c, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _chk(g.contiguous(), "grad_cost")
        b, c, h, w = ctx.dims
        shape = (b, h, w, c) if ctx.cl else (b, c, h, w)
        gl, gr = g.new_empty(shape), g.new_empty(shape)
        name = "az_cost_volume_bwd_ndhwc" if ctx.cl else "az_cost_volume_bwd"
        with torch.cuda.device(g.device):
            _call(name, _p(gl), _p(gr), _p(g), b, c, ctx.ndisp, h, w, _stream())
        return gl, gr, None, None


def cost_volume(feat_l, feat_r, ndisp):
    """[B,C,h,w] x2 -> [B,2C,ndisp,h,w] (reference layout)."""
    return _CostVolume.apply(feat_l, feat_r, int(ndisp), False)


def cost_volume_ndhwc(feat_l_nhwc, feat_r_nhwc, ndisp):
    """[B,h,w,C] x2 -> [B,ndisp,h,w,2C] (channels-last, internal layout)."""
    return _CostVolume.apply(feat_l_nhwc, feat_r_nhwc, int(ndisp), True)


# ----------------------------------------------------------------------------
# K6 soft-argmin head
# ----------------------------------------------------------------------------
class _SoftArgmin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        # logits: [B,1,d,h,w] or [B,d,h,w]
        lg = _chk(logits.contiguous(), "logits")
        if lg.dim() == 5:
            if lg.shape[1] != 1:
                raise RuntimeError("logits must have one channel")
            b, _, d, h, w = lg.shape
        else:
            b, d, h, w = lg.shape
        out = lg.new_empty(b, 1, 4 * h, 4 * w)
        # per-pixel softmax shift / normaliser (8 B per pixel) so that backward skips that pass
        stats = lg.new_empty(b, 4 * h, 4 * w, 2) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(lg.device):
            _call("az_softargmin_fwd", _p(out), _p(stats), _p(lg), b, d, h, w, _stream())
        if stats is not None:
            ctx.save_for_backward(lg, stats, out)
        else:
            ctx.save_for_backward(lg)
        ctx.dims = (b, d, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        lg, stats, out = (tuple(ctx.saved_tensors) + (None, None))[:3]
        g = _chk(g.contiguous(), "grad_disp")
        b, d, h, w = ctx.dims
        gl = torch.empty_like(lg)
        with torch.cuda.device(g.device):
            _call("az_softargmin_bwd", _p(gl), _p(g), _p(lg), _p(stats), _p(out), b, d, h, w, _stream())
        return gl


def softargmin(logits):
    """Fused trilinear x4 upsample + softmax over D=4d + expectation -> [B,1,4h,4w]."""
    return _SoftArgmin.apply(logits)


# ----------------------------------------------------------------------------
# K7 bilinear gather warp
# ----------------------------------------------------------------------------
class _WarpGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, disp):
        img = _chk(img.contiguous(), "img")
        disp = _chk(disp.contiguous(), "disp")
        b, c, h, w = img.shape
        if disp.numel() != b * h * w:
            raise RuntimeError("disp must be [B,1,H,W]")
        out = torch.empty_like(img)
        with torch.cuda.device(img.device):
            _call("az_warp_gather_fwd", _p(out), _p(img), _p(disp), b, c, h, w, _stream())
        ctx.save_for_backward(img, disp)
        return out

    @staticmethod
    def backward(ctx, g):
        img, disp = ctx.saved_tensors
        g = _chk(g.contiguous(), "grad_out")
        b, c, h, w = img.shape
        gd = torch.empty_like(disp)
        gi = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(g.device):
            _call("az_warp_gather_bwd", _p(gd), _p(gi), _p(g), _p(img), _p(disp), b, c, h, w, _stream())
        return gi, gd


def warp_gather(img, disp):
    """apply_disparity: bilinear sample of img at x + disp, zero padding."""
    return _WarpGather.apply(img, disp)


# ----------------------------------------------------------------------------
# K8 fused patch reprojection loss
# ----------------------------------------------------------------------------
class _PatchReproj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pat_l, pat_r, disp, mask_u8, ps, sign):
        b, c, h, w = pat_l.shape
        acc = torch.empty(2, dtype=torch.float64, device=pat_l.device)
        with torch.cuda.device(pat_l.device):
            _call("az_patch_reproj_fwd", _p(acc), _p(pat_l), _p(pat_r), _p(disp), _p(mask_u8),
                  b, c, h, w, ps, float(sign), _stream())
        ctx.save_for_backward(pat_l, pat_r, disp, mask_u8, acc)
        ctx.ps, ctx.sign = ps, sign
        return (acc[0] / acc[1]).to(torch.float32)

    @staticmethod
    def backward(ctx, gloss):
        pat_l, pat_r, disp, mask_u8, acc = ctx.saved_tensors
        b, c, h, w = pat_l.shape
        gloss = gloss.to(torch.float32).contiguous()
        gd = torch.empty_like(disp)
        with torch.cuda.device(disp.device):
            _call("az_patch_reproj_bwd", _p(gd), _p(gloss), _p(acc), _p(pat_l), _p(pat_r),
                  _p(disp), _p(mask_u8), b, c, h, w, ctx.ps, float(ctx.sign), _stream())
        return None, None, gd, None, None, None


def patch_reprojection(input_l, input_r, pred_disp_l, mask=None, ps=5, want_vis=True):
    """get_reproj_error_patch -> (loss, warped_vis [B,C,H,W], mask [B,C,H,W] int32)."""
    pat_l = _chk(input_l.detach().contiguous(), "input_L")
    pat_r = _chk(input_r.detach().contiguous(), "input_R")
    disp = _chk(pred_disp_l.contiguous(), "pred_disp_l")
    b, c, h, w = pat_l.shape
    if disp.numel() != b * h * w:
        raise RuntimeError("pred_disp_l must be [B,1,H,W]")
    mask_u8 = None
    if mask is not None:
        if mask.numel() != b * h * w:
            raise RuntimeError("mask must be [B,1,H,W]")
        mask_u8 = _chk(mask.contiguous().to(torch.uint8), "mask", torch.uint8)
    loss = _PatchReproj.apply(pat_l, pat_r, disp, mask_u8, int(ps), -1.0)
    vis = None
    if want_vis:
        vis = torch.empty_like(pat_l)
        with torch.cuda.device(pat_l.device):
            _call("az_patch_reproj_vis", _p(vis), _p(pat_r), _p(disp.detach()), b, c, h, w, int(ps),
                  -1.0, _stream())
    if mask is None:
        mask_out = torch.ones(b, c, h, w, dtype=torch.int32, device=pat_l.device)
    else:
        mask_out = mask.reshape(b, 1, h, w).expand(b, c, h, w).to(torch.int32)
    return loss, vis, mask_out


# ----------------------------------------------------------------------------
# K9 local contrast normalisation
# ----------------------------------------------------------------------------
def local_contrast_norm(image, kernel_size=9, eps=1e-5):
    """image [B,1,H,W] (contiguous) -> (normed, std), both [B,1,H,W]."""
    img = _chk(image, "image")
    b, c, h, w = img.shape
    normed = img.new_empty(b, 1, h, w)
    std = img.new_empty(b, 1, h, w)
    with torch.cuda.device(img.device):
        _call("az_lcn", _p(normed), _p(std), _p(img), b, h, w, int(kernel_size), float(eps),
              c * h * w, _stream())
    return normed, std


# ----------------------------------------------------------------------------
# K12 disparity loss + error metrics (the step after the path)
# ----------------------------------------------------------------------------
def _mask_u8(mask, like, name="mask"):
    if mask is None:
        return None
    if mask.shape != like.shape:
        raise RuntimeError(f"{name}: shape {tuple(mask.shape)} != {tuple(like.shape)}")
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)  # same bytes, no copy
    return _chk(mask, name, torch.uint8)


class _DispLoss(torch.autograd.Function):
    WEIGHTS = (1.0, 0.7, 0.5)  # pred3, pred2, pred1 (utils/losses.py:9-14)

    @staticmethod
    def forward(ctx, pred3, pred2, pred1, gt, mask, lo, hi):
        p3, p2, p1 = (_chk(p.contiguous(), f"pred{k}") for p, k in ((pred3, 3), (pred2, 2), (pred1, 1)))
        gt = _chk(gt.contiguous(), "disp_gt")
        if not (p3.shape == p2.shape == p1.shape == gt.shape):
            raise RuntimeError("predictions and ground truth must have identical shapes")
        m = _mask_u8(mask, gt)
        acc = torch.zeros(4, dtype=torch.float64, device=gt.device)
        with torch.cuda.device(gt.device):
            _call("az_disp_loss_fwd", _p(acc), _p(p3), _p(p2), _p(p1), _p(gt), _p(m), float(lo), float(hi),
                  gt.numel(), _stream())
        w3, w2, w1 = _DispLoss.WEIGHTS
        loss = ((w3 * acc[0] + w2 * acc[1] + w1 * acc[2]) / acc[3]).to(torch.float32)  # 0/0 = nan, as the reference's empty mean
        ctx.save_for_backward(p3, p2, p1, gt, m if m is not None else gt.new_empty(0), acc)
        ctx.has_mask, ctx.lo, ctx.hi = m is not None, float(lo), float(hi)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p3, p2, p1, gt, m, acc = ctx.saved_tensors
        gloss = _chk(gloss.contiguous().to(torch.float32).reshape(1), "grad_loss")
        g3, g2, g1 = torch.empty_like(p3), torch.empty_like(p2), torch.empty_like(p1)
        w3, w2, w1 = _DispLoss.WEIGHTS
        with torch.cuda.device(gt.device):
            _call("az_disp_loss_bwd", _p(g3), _p(g2), _p(g1), _p(p3), _p(p2), _p(p1), _p(gt),
                  _p(m) if ctx.has_mask else None, ctx.lo, ctx.hi, _p(gloss), _p(acc), w3, w2, w1,
                  gt.numel(), _stream())
        return g3, g2, g1, None, None, None, None


def disp_loss(pred3, pred2, pred1, gt, mask=None, lo=0.0, hi=float("inf")):
    """smooth_l1(pred3) + 0.7 smooth_l1(pred2) + 0.5 smooth_l1(pred1), each a mean over the valid
    pixels: `mask` (bool/uint8, same shape) or, when mask is None, lo < gt < hi.  One pass, no sync."""
    return _DispLoss.apply(pred3, pred2, pred1, gt, mask, lo, hi)


def disp_metrics(disp_gt, depth_gt, disp_pred, mask, focal_x_baseline=None, depth_pred=None):
    """fp64 sums [sum|dd|, #(|dd|>1), #(|dd|>2), sum clip(|1000 dz|,0,100), #(|dz|>2e-3), #(|dz|>4e-3),
    #(|dz|>8e-3), count] over the masked pixels, on the device (no sync)."""
    dg = _chk(disp_gt.contiguous(), "disp_gt")
    zg = _chk(depth_gt.contiguous(), "depth_gt")
    dp = _chk(disp_pred.contiguous(), "disp_pred")
    if not (dg.shape == zg.shape == dp.shape):
        raise RuntimeError("disp_gt, depth_gt and disp_pred must have identical shapes")
    m = _mask_u8(mask, dg)
    if m is None:
        raise RuntimeError("mask is required")
    b = dg.shape[0]
    zp = _chk(depth_pred.contiguous(), "depth_pred") if depth_pred is not None else None
    fb = None
    if zp is None:
        fb = _chk(focal_x_baseline.contiguous().reshape(-1), "focal_x_baseline")
        if fb.numel() != b:
            raise RuntimeError("focal_x_baseline must hold one value per batch element")
    acc = torch.zeros(8, dtype=torch.float64, device=dg.device)
    with torch.cuda.device(dg.device):
        _call("az_disp_metrics", _p(acc), _p(dg), _p(zg), _p(dp), _p(zp), _p(fb), _p(m), b,
              dg.numel() // max(b, 1), _stream())
    return acc
