"""torch.library registration of the path's API-surface operators as `azhip::*`: the same C-ABI
calls as activezero_amd/ops.py, exposed through PyTorch's dispatcher so that they are visible to
torch.compile / FakeTensor tracing (shape inference runs on the meta device, without a GPU) and carry
registered autograd formulas instead of Python autograd.Function objects.

    torch.ops.azhip.warp_scatter(img, disp_int32, sign)            K1/K2   utils/warp_ops.py:55-95
    torch.ops.azhip.cost_volume(feat_l, feat_r, ndisp)             K3      nets/psmnet/psmnet_3.py:149-163
    torch.ops.azhip.softargmin(logits)                             K6      nets/psmnet/psmnet_3.py:184-215
    torch.ops.azhip.warp_gather(img, disp)                         K7      utils/reprojection.py:13-35
    torch.ops.azhip.local_contrast_norm(image, kernel_size, eps)   K9      utils/reprojection.py:175-200

Importing this module registers the operators (idempotent).  There is no CPU implementation: calling one
with CPU tensors raises, as every other entry of this package does.
"""
import torch

from . import ops
from .ops import _call, _chk, _p, _stream

_LIB = torch.library.Library("azhip", "DEF")
_LIB.define("warp_scatter(Tensor img, Tensor disp, int sign) -> Tensor")
_LIB.define("cost_volume(Tensor feat_l, Tensor feat_r, int ndisp) -> Tensor")
_LIB.define("cost_volume_bwd(Tensor grad_cost, int ndisp) -> (Tensor, Tensor)")
_LIB.define("softargmin_fwd(Tensor logits) -> (Tensor, Tensor)")
_LIB.define("softargmin_bwd(Tensor grad_disp, Tensor logits, Tensor stats, Tensor disp) -> Tensor")
_LIB.define("softargmin(Tensor logits) -> Tensor")
_LIB.define("warp_gather(Tensor img, Tensor disp) -> Tensor")
_LIB.define("warp_gather_bwd(Tensor grad_out, Tensor img, Tensor disp, bool need_img_grad) -> (Tensor, Tensor)")
_LIB.define("local_contrast_norm(Tensor image, int kernel_size, float eps) -> (Tensor, Tensor)")


def _no_cpu(name):
    def impl(*_a, **_k):
        raise RuntimeError(f"azhip::{name}: tensors must live on the GPU (the HIP path has no CPU fallback)")
    return impl


# ---- K1/K2 ------------------------------------------------------------------------------------------
def _warp_scatter(img, disp, sign):
    return ops.warp_scatter(img, disp, sign)


_LIB.impl("warp_scatter", _warp_scatter, "CUDA")
_LIB.impl("warp_scatter", _no_cpu("warp_scatter"), "CPU")
torch.library.register_fake("azhip::warp_scatter", lambda img, disp, sign: torch.empty_like(img))


# ---- K3 ---------------------------------------------------------------------------------------------
def _cost_volume(feat_l, feat_r, ndisp):
    fl, fr = _chk(feat_l.contiguous(), "feat_l"), _chk(feat_r.contiguous(), "feat_r")
    if fl.shape != fr.shape:
        raise RuntimeError("feature maps must have identical shapes")
    b, c, h, w = fl.shape
    out = fl.new_empty(b, 2 * c, ndisp, h, w)
    with torch.cuda.device(fl.device):
        _call("az_cost_volume_fwd", _p(out), _p(fl), _p(fr), b, c, ndisp, h, w, _stream())
    return out


def _cost_volume_bwd(g, ndisp):
    g = _chk(g.contiguous(), "grad_cost")
    b, c2, d, h, w = g.shape
    gl, gr = g.new_empty(b, c2 // 2, h, w), g.new_empty(b, c2 // 2, h, w)
    with torch.cuda.device(g.device):
        _call("az_cost_volume_bwd", _p(gl), _p(gr), _p(g), b, c2 // 2, ndisp, h, w, _stream())
    return gl, gr


_LIB.impl("cost_volume", _cost_volume, "CUDA")
_LIB.impl("cost_volume", _no_cpu("cost_volume"), "CPU")
_LIB.impl("cost_volume_bwd", _cost_volume_bwd, "CUDA")
torch.library.register_fake("azhip::cost_volume",
                            lambda fl, fr, nd: fl.new_empty(fl.shape[0], 2 * fl.shape[1], nd, fl.shape[2], fl.shape[3]))
torch.library.register_fake("azhip::cost_volume_bwd",
                            lambda g, nd: (g.new_empty(g.shape[0], g.shape[1] // 2, g.shape[3], g.shape[4]),
                                           g.new_empty(g.shape[0], g.shape[1] // 2, g.shape[3], g.shape[4])))
torch.library.register_autograd(
    "azhip::cost_volume", lambda ctx, g: (*torch.ops.azhip.cost_volume_bwd(g, ctx.ndisp), None),
    setup_context=lambda ctx, inputs, output: setattr(ctx, "ndisp", inputs[2]))


# ---- K6 ---------------------------------------------------------------------------------------------
def _dims4(lg):
    if lg.dim() == 5:
        if lg.shape[1] != 1:
            raise RuntimeError("logits must have one channel")
        return lg.shape[0], lg.shape[2], lg.shape[3], lg.shape[4]
    return tuple(lg.shape)


def _softargmin_fwd(logits):
    lg = _chk(logits.contiguous(), "logits")
    b, d, h, w = _dims4(lg)
    out = lg.new_empty(b, 1, 4 * h, 4 * w)
    stats = lg.new_empty(b, 4 * h, 4 * w, 2)
    with torch.cuda.device(lg.device):
        _call("az_softargmin_fwd", _p(out), _p(stats), _p(lg), b, d, h, w, _stream())
    return out, stats


def _softargmin_bwd(g, logits, stats, disp):
    lg = _chk(logits.contiguous(), "logits")
    g = _chk(g.contiguous(), "grad_disp")
    b, d, h, w = _dims4(lg)
    gl = torch.empty_like(lg)
    with torch.cuda.device(g.device):
        _call("az_softargmin_bwd", _p(gl), _p(g), _p(lg), _p(stats), _p(disp), b, d, h, w, _stream())
    return gl


def _softargmin_fake(lg):
    b, d, h, w = _dims4(lg)
    return lg.new_empty(b, 1, 4 * h, 4 * w), lg.new_empty(b, 4 * h, 4 * w, 2)


_LIB.impl("softargmin_fwd", _softargmin_fwd, "CUDA")
_LIB.impl("softargmin_fwd", _no_cpu("softargmin"), "CPU")
_LIB.impl("softargmin_bwd", _softargmin_bwd, "CUDA")
_LIB.impl("softargmin", lambda lg: torch.ops.azhip.softargmin_fwd(lg)[0], "CompositeImplicitAutograd")
torch.library.register_fake("azhip::softargmin_fwd", _softargmin_fake)
torch.library.register_fake("azhip::softargmin_bwd", lambda g, lg, st, dp: torch.empty_like(lg))


def _sa_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1], output[0])


def _sa_backward(ctx, g_out, g_stats):
    lg, stats, disp = ctx.saved_tensors
    return torch.ops.azhip.softargmin_bwd(g_out, lg, stats, disp)


torch.library.register_autograd("azhip::softargmin_fwd", _sa_backward, setup_context=_sa_setup)


# ---- K7 ---------------------------------------------------------------------------------------------
def _warp_gather(img, disp):
    img, disp = _chk(img.contiguous(), "img"), _chk(disp.contiguous(), "disp")
    b, c, h, w = img.shape
    if disp.numel() != b * h * w:
        raise RuntimeError("disp must be [B,1,H,W]")
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _call("az_warp_gather_fwd", _p(out), _p(img), _p(disp), b, c, h, w, _stream())
    return out


def _warp_gather_bwd(g, img, disp, need_img_grad):
    g = _chk(g.contiguous(), "grad_out")
    b, c, h, w = img.shape
    gd = torch.empty_like(disp)
    gi = torch.zeros_like(img) if need_img_grad else img.new_empty(0)
    with torch.cuda.device(g.device):
        _call("az_warp_gather_bwd", _p(gd), _p(gi) if need_img_grad else None, _p(g), _p(img.contiguous()),
              _p(disp.contiguous()), b, c, h, w, _stream())
    return gd, gi


_LIB.impl("warp_gather", _warp_gather, "CUDA")
_LIB.impl("warp_gather", _no_cpu("warp_gather"), "CPU")
_LIB.impl("warp_gather_bwd", _warp_gather_bwd, "CUDA")
torch.library.register_fake("azhip::warp_gather", lambda img, disp: torch.empty_like(img))
torch.library.register_fake("azhip::warp_gather_bwd",
                            lambda g, img, disp, need: (torch.empty_like(disp), torch.empty_like(img) if need else img.new_empty(0)))


def _wg_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _wg_backward(ctx, g):
    img, disp = ctx.saved_tensors
    need = ctx.needs_input_grad[0]
    gd, gi = torch.ops.azhip.warp_gather_bwd(g, img, disp, need)
    return (gi if need else None), gd


torch.library.register_autograd("azhip::warp_gather", _wg_backward, setup_context=_wg_setup)


# ---- K9 ---------------------------------------------------------------------------------------------
_LIB.impl("local_contrast_norm", lambda image, k, eps: ops.local_contrast_norm(image.contiguous(), k, eps), "CUDA")
_LIB.impl("local_contrast_norm", _no_cpu("local_contrast_norm"), "CPU")
torch.library.register_fake(
    "azhip::local_contrast_norm",
    lambda image, k, eps: (image.new_empty(image.shape[0], 1, image.shape[2], image.shape[3]),
                           image.new_empty(image.shape[0], 1, image.shape[2], image.shape[3])))
