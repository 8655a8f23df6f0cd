"""torch.library registration of the path's API-surface operators as `azhip::*`: the same C-ABI
calls as activezero_amd/ops.py, exposed through PyTorch's dispatcher so that they are visible to
torch.compile / FakeTensor tracing (shape inference runs on the meta device, without a GPU) and carry
registered autograd formulas instead of Python autograd.Function objects.

    torch.ops.azhip.warp_scatter(img, disp_int32, sign)            K1/K2   utils/warp_ops.py:55-95
    torch.ops.azhip.cost_volume(feat_l, feat_r, ndisp)             K3      nets/psmnet/psmnet_3.py:149-163
    torch.ops.azhip.softargmin(logits)                             K6      nets/psmnet/psmnet_3.py:184-215
    torch.ops.azhip.warp_gather(img, disp)                         K7      utils/reprojection.py:13-35
    torch.ops.azhip.local_contrast_norm(image, kernel_size, eps)   K9      utils/reprojection.py:175-200
    torch.ops.azhip.conv3d(x_ndhwc, weight, mode)                  K4/K5   nets/psmnet/psmnet_3.py:15-58 (Conv3d /
                                                                           ConvTranspose3d 3x3x3; mode 0 stride 1, 1 stride 2,
                                                                           2 transposed stride 2; channels-last volume)
    torch.ops.azhip.bn3d(raw, scale, shift, residual?, relu)       --      psmnet_submodule_3.py:44-56 (the affine form of
                                                                           BatchNorm3d (+ residual) (+ ReLU) on a conv output)
    torch.ops.azhip.patch_reproj(left, right, disp, mask?, ps)     K8      utils/reprojection.py:81-172

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


# ---- K4/K5: 3x3x3 convolution of a channels-last volume ------------------------------------------------
from . import conv3d as _c3  # noqa: E402

_LIB.define("conv3d(Tensor x, Tensor weight, int mode) -> Tensor")
_LIB.define("conv3d_input_grad(Tensor grad_out, Tensor weight, int mode) -> Tensor")
_LIB.define("conv3d_weight_grad(Tensor x, Tensor grad_out, Tensor weight, int mode) -> Tensor")
_LIB.define("bn3d(Tensor raw, Tensor scale, Tensor shift, Tensor? residual, bool relu) -> Tensor")
_LIB.define("patch_reproj(Tensor left, Tensor right, Tensor disp, Tensor? mask, int ps) -> Tensor")
_LIB.define("patch_reproj_bwd(Tensor grad_loss, Tensor left, Tensor right, Tensor disp, Tensor? mask, int ps) -> Tensor")


def _c3_channels(weight, mode):
    return (weight.shape[0], weight.shape[1]) if mode == _c3.DECONV_S2 else (weight.shape[1], weight.shape[0])


def _conv3d(x, weight, mode):
    with torch.cuda.device(x.device):
        return _c3._conv(_chk(x.contiguous(), "x"), weight, mode, _c3.DEFAULT_ARITH.conv)


def _conv3d_input_grad(g, weight, mode):
    cin, cout = _c3_channels(weight, mode)
    a = _c3.DEFAULT_ARITH
    with torch.cuda.device(g.device):
        return _c3._input_grad(_chk(g.contiguous(), "grad_out"), weight, mode, cin, cout, _c3.F16X3 if a.bwd16 else a.conv)


def _conv3d_weight_grad(x, g, weight, mode):
    cin, cout = _c3_channels(weight, mode)
    a = _c3.DEFAULT_ARITH
    with torch.cuda.device(g.device):
        return _c3._weight_grad(_chk(x.contiguous(), "x"), _chk(g.contiguous(), "grad_out"), mode, cin, cout,
                                _c3.F16X3 if a.bwd16 else a.wgrad)


def _conv3d_fake(x, weight, mode):
    b, d, h, w, _ = x.shape
    do, ho, wo = _c3._out_dims(mode, d, h, w)
    return x.new_empty(b, do, ho, wo, weight.shape[1] if mode == _c3.DECONV_S2 else weight.shape[0])


def _conv3d_in_dims(g, weight, mode):
    b, d, h, w, _ = g.shape
    cin, _ = _c3_channels(weight, mode)
    if mode == _c3.CONV_S1:
        return g.new_empty(b, d, h, w, cin)
    if mode == _c3.CONV_S2:
        return g.new_empty(b, 2 * d, 2 * h, 2 * w, cin)
    return g.new_empty(b, d // 2, h // 2, w // 2, cin)


for _n, _f in (("conv3d", _conv3d), ("conv3d_input_grad", _conv3d_input_grad), ("conv3d_weight_grad", _conv3d_weight_grad)):
    _LIB.impl(_n, _f, "CUDA")
    _LIB.impl(_n, _no_cpu(_n), "CPU")
torch.library.register_fake("azhip::conv3d", _conv3d_fake)
torch.library.register_fake("azhip::conv3d_input_grad", _conv3d_in_dims)
torch.library.register_fake("azhip::conv3d_weight_grad", lambda x, g, w, mode: torch.empty_like(w))


def _conv3d_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.mode = inputs[2]


def _conv3d_backward(ctx, g):
    x, w = ctx.saved_tensors
    gx = torch.ops.azhip.conv3d_input_grad(g, w, ctx.mode) if ctx.needs_input_grad[0] else None
    gw = torch.ops.azhip.conv3d_weight_grad(x, g, w, ctx.mode) if ctx.needs_input_grad[1] else None
    return gx, gw, None


torch.library.register_autograd("azhip::conv3d", _conv3d_backward, setup_context=_conv3d_setup)


# ---- BatchNorm3d as its affine map (+ residual) (+ ReLU) on a channels-last tensor ------------------------
def _bn3d(raw, scale, shift, residual, relu):
    raw = _chk(raw.contiguous(), "raw")
    c = raw.shape[-1]
    y = torch.empty_like(raw)
    with torch.cuda.device(raw.device):
        _call("az_bn3d_apply", _p(y), _p(raw), _p(_chk(scale.contiguous(), "scale")), _p(_chk(shift.contiguous(), "shift")),
              _p(_chk(residual.contiguous(), "residual")) if residual is not None else None, int(relu),
              raw.numel() // c, c, None, _stream())
    return y


_LIB.impl("bn3d", _bn3d, "CUDA")
_LIB.impl("bn3d", _no_cpu("bn3d"), "CPU")
torch.library.register_fake("azhip::bn3d", lambda raw, scale, shift, residual, relu: torch.empty_like(raw))


def _bn3d_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output)
    ctx.relu, ctx.has_res = inputs[4], inputs[3] is not None


def _bn3d_backward(ctx, g):
    raw, scale, y = ctx.saved_tensors
    dz = g * (y > 0).to(g.dtype) if ctx.relu else g
    red = tuple(range(raw.dim() - 1))
    return dz * scale, (dz * raw).sum(dim=red), dz.sum(dim=red), (dz if ctx.has_res else None), None


torch.library.register_autograd("azhip::bn3d", _bn3d_backward, setup_context=_bn3d_setup)


# ---- K8: patch reprojection loss --------------------------------------------------------------------------
def _pr_args(left, right, disp, mask):
    pl, pr = _chk(left.detach().contiguous(), "left"), _chk(right.detach().contiguous(), "right")
    d = _chk(disp.contiguous(), "disp")
    b, c, h, w = pl.shape
    if d.numel() != b * h * w:
        raise RuntimeError("disp must be [B,1,H,W]")
    m = _chk(mask.contiguous().to(torch.uint8), "mask", torch.uint8) if mask is not None else None
    return pl, pr, d, m, (b, c, h, w)


def _patch_reproj(left, right, disp, mask, ps):
    pl, pr, d, m, (b, c, h, w) = _pr_args(left, right, disp, mask)
    acc = torch.empty(2, dtype=torch.float64, device=pl.device)
    with torch.cuda.device(pl.device):
        _call("az_patch_reproj_fwd", _p(acc), _p(pl), _p(pr), _p(d), _p(m), b, c, h, w, int(ps), -1.0, _stream())
    return (acc[0] / acc[1]).to(torch.float32)


def _patch_reproj_bwd(gloss, left, right, disp, mask, ps):
    pl, pr, d, m, (b, c, h, w) = _pr_args(left, right, disp, mask)
    acc = torch.empty(2, dtype=torch.float64, device=pl.device)
    gd = torch.empty_like(d)
    with torch.cuda.device(pl.device):  # (the normaliser is recomputed: the dispatcher form saves tensors only)
        _call("az_patch_reproj_fwd", _p(acc), _p(pl), _p(pr), _p(d), _p(m), b, c, h, w, int(ps), -1.0, _stream())
        _call("az_patch_reproj_bwd", _p(gd), _p(gloss.to(torch.float32).contiguous()), _p(acc), _p(pl), _p(pr), _p(d), _p(m),
              b, c, h, w, int(ps), -1.0, _stream())
    return gd


_LIB.impl("patch_reproj", _patch_reproj, "CUDA")
_LIB.impl("patch_reproj", _no_cpu("patch_reproj"), "CPU")
_LIB.impl("patch_reproj_bwd", _patch_reproj_bwd, "CUDA")
torch.library.register_fake("azhip::patch_reproj", lambda l, r, d, m, ps: l.new_empty(()))
torch.library.register_fake("azhip::patch_reproj_bwd", lambda g, l, r, d, m, ps: torch.empty_like(d))


def _pr_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2], *( [inputs[3]] if inputs[3] is not None else []))
    ctx.has_mask, ctx.ps = inputs[3] is not None, inputs[4]


def _pr_backward(ctx, g):
    left, right, disp, *m = ctx.saved_tensors
    gd = torch.ops.azhip.patch_reproj_bwd(g, left, right, disp, m[0] if ctx.has_mask else None, ctx.ps)
    return None, None, gd, None, None


torch.library.register_autograd("azhip::patch_reproj", _pr_backward, setup_context=_pr_setup)
