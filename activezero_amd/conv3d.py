"""Autograd wrappers of the MFMA 3-D convolution family (include/azhip.h K4/K5).

Volumes are channels-last tensors [B, D, H, W, C] (C in {32, 64}); weights stay in
PyTorch's native layouts ([Cout,Cin,3,3,3] for Conv3d, [Cin,Cout,3,3,3] for
ConvTranspose3d) so the reference state-dict loads unchanged -- the kernels read a
packed copy made per call by az_conv3d_pack_weights (110-442 KB, L2 resident).

Every layer's three passes map onto the same gather kernel (mode 0/1/2 = stride-1
conv, stride-2 conv, stride-2 transposed conv) with differently packed weights:

    layer                 forward      input gradient            weight gradient
    Conv3d stride 1       mode 0       mode 0, flipped+swapped   wgrad(dy, x, 1)
    Conv3d stride 2       mode 1       mode 2, swapped           wgrad(dy, x, 2)
    ConvTranspose3d s2    mode 2       mode 1, as stored         wgrad(x, dy, 2)
"""
import os

import torch

from . import _lib, profiler
from .ops import _call, _chk, _p, _stream

CONV_S1, CONV_S2, DECONV_S2 = 0, 1, 2
FP32, BF16X6 = 0, 1
# arithmetic of the gather kernels (forward, input gradients, transposed convs):
#   fp32   -- v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chain (157 TFLOP/s peak)
#   bf16x6 -- exact 3-way bf16 split of both operands, six bf16 MFMAs per product, fp32
#             accumulate: fp32-class accuracy (measured ~1e-7 relative) at 2.7x the rate
PRECISION = {"fp32": FP32, "bf16x6": BF16X6}[os.environ.get("AZ_CONV_PRECISION", "bf16x6")]


WGRAD_PRECISION = {"fp32": FP32, "bf16x6": BF16X6}[os.environ.get("AZ_WGRAD_PRECISION", "bf16x6")]


def set_precision(name, wgrad=None):
    global PRECISION, WGRAD_PRECISION
    PRECISION = {"fp32": FP32, "bf16x6": BF16X6}[name]
    if wgrad is not None:
        WGRAD_PRECISION = {"fp32": FP32, "bf16x6": BF16X6}[wgrad]


def _dims(x):
    b, d, h, w, c = x.shape
    return b, d, h, w, c


def _out_dims(mode, d, h, w):
    if mode == CONV_S1:
        return d, h, w
    if mode == CONV_S2:
        return (d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return 2 * d, 2 * h, 2 * w


# Inference-time caches (used only while autograd is off): packed weights and folded BatchNorm
# affine maps are functions of parameters that do not change between forward passes.  Entries
# are keyed on the tensors' storage address AND version counter, so an optimizer step or a
# load_state_dict (both write in place) invalidates them; each entry also holds its source tensors,
# so their addresses cannot be recycled for other data while the entry lives.
_PACK_CACHE, _AFFINE_CACHE = {}, {}


def _cache_key(*tensors):
    return tuple((t.data_ptr(), t._version, t.device.index) for t in tensors)


def _inference_mode():
    """autograd off AND not inside a backward pass (autograd also switches grad mode off while it runs
    Function.backward; caching there would only churn: the weights change every step)."""
    return not torch.is_grad_enabled() and torch._C._current_graph_task_id() == -1


def _pack(weight, op_cin, op_cout, stride_out, stride_in, flip):
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if _inference_mode():
        key = (_cache_key(weight), op_cin, op_cout, stride_out, stride_in, bool(flip), PRECISION)
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            return hit[0]
    n = _lib.lib().az_conv3d_packed_floats(op_cin, op_cout, PRECISION)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _call("az_conv3d_pack_weights", _p(packed), _p(w), op_cin, op_cout, stride_out, stride_in,
          int(flip), PRECISION, _stream())
    if key is not None:
        if len(_PACK_CACHE) > 256:
            _PACK_CACHE.clear()
        _PACK_CACHE[key] = (packed, weight)
    return packed


def eval_affine(bn, like):
    """(scale, shift) of an eval-mode BatchNorm: gamma/sqrt(running_var+eps), beta - running_mean*scale."""
    c = bn.num_features
    key = None
    if _inference_mode():
        key = (_cache_key(bn.weight, bn.bias, bn.running_mean, bn.running_var), float(bn.eps))
        hit = _AFFINE_CACHE.get(key)
        if hit is not None:
            return hit[0], hit[1]
    scale, shift = like.new_empty(c), like.new_empty(c)
    _call("az_bn3d_eval_affine", _p(scale), _p(shift), _p(bn.weight.detach()), _p(bn.bias.detach()),
          _p(bn.running_mean), _p(bn.running_var), float(bn.eps), c, _stream())
    if key is not None:
        if len(_AFFINE_CACHE) > 512:
            _AFFINE_CACHE.clear()
        _AFFINE_CACHE[key] = (scale, shift, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    return scale, shift


def _pack_forward(weight, mode):
    if mode == DECONV_S2:  # [Cin, Cout, 27]
        cin, cout = weight.shape[0], weight.shape[1]
        return _pack(weight, cin, cout, 27, cout * 27, False), cin, cout
    cout, cin = weight.shape[0], weight.shape[1]
    return _pack(weight, cin, cout, cin * 27, 27, False), cin, cout


def _peak(precision):
    """MFMA roofline of the arithmetic in use, in fp32-equivalent TFLOP/s: the fp32 MFMA dense
    peak, or the bf16 dense peak divided by the six bf16 MFMAs one fp32 product costs."""
    return (157.3, "fp32 MFMA") if precision == FP32 else (2500.0 / 6.0, "bf16x6: bf16 MFMA peak / 6")


def _conv_flops(b, vox_out, cin, cout, mode):
    taps = 27.0 / 8.0 if mode == DECONV_S2 else 27.0
    return 2.0 * taps * cin * cout * b * vox_out


class LazyCostVolume:
    """The PSMNet concat cost volume as a recipe (left/right NHWC features + number of
    disparity planes) instead of a [B,d,h,w,64] tensor: the first convolution
    synthesises its operand on the fly (az_conv3d_fwd, src = 1)."""

    def __init__(self, feat_l_nhwc, feat_r_nhwc, ndisp):
        self.fl, self.fr, self.ndisp = _chk(feat_l_nhwc, "feat_l"), _chk(feat_r_nhwc, "feat_r"), int(ndisp)
        if self.fl.shape != self.fr.shape or self.fl.shape[-1] != 32:
            raise RuntimeError("fused cost volume expects two [B,h,w,32] feature maps")


def _run_gather(x, packed, mode, cin, cout, scale=None, shift=None, residual=None, relu=False,
                stats=False, tag="conv3d"):
    if isinstance(x, LazyCostVolume):
        if stats or mode != CONV_S1 or cin != 64:
            raise RuntimeError("the fused cost-volume operand is only wired for the eval-mode dres0[0] conv")
        b, h, w, _ = x.fl.shape
        d = x.ndisp
        out = x.fl.new_empty(b, d, h, w, cout)
        with profiler.scope(f"{tag}_costvol_m0_{cin}_{cout}", flops=_conv_flops(b, d * h * w, cin, cout, mode),
                            peak=_peak(PRECISION)):
            _call("az_conv3d_fwd", _p(out), _p(x.fl), _p(x.fr), _p(packed), _p(scale), _p(shift),
                  _p(residual), int(relu), mode, 1, PRECISION, b, cin, cout, d, h, w, _stream())
        return out
    b, d, h, w, c = _dims(x)
    assert c == cin, (c, cin)
    if mode == CONV_S2 and (d % 2 or h % 2 or w % 2):
        raise RuntimeError("stride-2 layers need even D/H/W (as PSMNet's hourglass does)")
    do, ho, wo = _out_dims(mode, d, h, w)
    out = x.new_empty(b, do, ho, wo, cout)
    flops = _conv_flops(b, do * ho * wo, cin, cout, mode)
    if d == 1 and mode == CONV_S1:
        flops /= 3.0  # a depth-1 volume (2-D layer): only the 9 taps of the centre depth slice exist
    name = f"{tag}_m{mode}_{cin}_{cout}"
    if stats:
        ntiles = _lib.lib().az_conv3d_num_tiles(mode, b, d, h, w)
        part = x.new_empty(cout, ntiles, 2)
        cnt = x.new_empty(ntiles)
        with profiler.scope(name, flops=flops, peak=_peak(PRECISION)):
            _call("az_conv3d_fwd_stats", _p(out), _p(part), _p(cnt), _p(x), None, _p(packed), mode, 0,
                  PRECISION, b, cin, cout, d, h, w, _stream())
        return out, part, cnt, ntiles
    with profiler.scope(name, flops=flops, peak=_peak(PRECISION)):
        _call("az_conv3d_fwd", _p(out), _p(x), None, _p(packed), _p(scale), _p(shift), _p(residual),
              int(relu), mode, 0, PRECISION, b, cin, cout, d, h, w, _stream())
    return out


def _wgrad(coarse, fine, stride, cm, cn, tag):
    b, dc, hc, wc, _ = _dims(coarse)
    _, df, hf, wf, _ = _dims(fine)
    gw = coarse.new_empty(cm, cn, 3, 3, 3)
    ws_bytes = _lib.lib().az_conv3d_wgrad_workspace(cm, cn)
    ws = coarse.new_empty(ws_bytes // 4)
    taps = 9 if (dc == 1 and df == 1) else 27  # depth-1 volumes: the 2-D layers
    with profiler.scope(f"{tag}_wgrad_s{stride}_{cm}_{cn}", flops=2.0 * taps * cm * cn * b * dc * hc * wc,
                        peak=_peak(WGRAD_PRECISION)):
        _call("az_conv3d_wgrad", _p(gw), _p(ws), ws_bytes, _p(coarse), _p(fine), stride, WGRAD_PRECISION,
              b, cm, cn, dc, hc, wc, df, hf, wf, _stream())
    return gw


class _ConvBN(torch.autograd.Function):
    """y = relu?( BN(conv(x, weight)) + residual ), one autograd node per convbn_3d unit."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, mode, relu, want_grad):
        x = _chk(x, "x")  # (LazyCostVolume never reaches autograd: see conv_bn)
        if residual is not None:
            residual = _chk(residual, "residual")
        training = bn.training or not bn.track_running_stats
        eps = float(bn.eps)
        with torch.cuda.device(x.device):
            packed, cin, cout = _pack_forward(weight, mode)
            if not training:
                scale, shift = eval_affine(bn, x)
                if want_grad and any(ctx.needs_input_grad):
                    raise NotImplementedError(
                        "eval-mode BatchNorm backward is not implemented on the HIP path; "
                        "run validation under torch.no_grad() as the reference does")
                return _run_gather(x, packed, mode, cin, cout, scale, shift, residual, relu)
            raw, part, cnt, ntiles = _run_gather(x, packed, mode, cin, cout, stats=True)
            scale, shift = x.new_empty(cout), x.new_empty(cout)
            mean, invstd = x.new_empty(cout), x.new_empty(cout)
            track = bn.track_running_stats and bn.running_mean is not None
            momentum = 0.1 if bn.momentum is None else float(bn.momentum)
            _call("az_bn3d_finalize", _p(mean), _p(invstd), _p(scale), _p(shift),
                  _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                  _p(part), _p(cnt), _p(gamma.detach()), _p(beta.detach()), ntiles, cout, eps,
                  momentum, _stream())
            if track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            y = torch.empty_like(raw)
            nvox = raw.numel() // cout
            with profiler.scope(f"bn3d_apply_{cout}", bytes=4.0 * raw.numel() * (3 if residual is not None else 2),
                                bound="hbm"):
                _call("az_bn3d_apply", _p(y), _p(raw), _p(scale), _p(shift), _p(residual), int(relu),
                      nvox, cout, _stream())
        # a ReLU layer without residual recomputes its mask from raw in backward (fma(raw, scale, shift) > 0)
        remask = relu and residual is None
        ctx.save_for_backward(x, weight, gamma, raw, y if (relu and not remask) else None, mean, invstd,
                              scale if remask else None, shift if remask else None)
        ctx.cfg = (mode, relu, residual is not None, cin, cout)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, gamma, raw, y, mean, invstd, scale, shift = ctx.saved_tensors
        mode, relu, has_res, cin, cout = ctx.cfg
        gy = _chk(gy.contiguous(), "grad_y")
        nvox = raw.numel() // cout
        with torch.cuda.device(gy.device):
            lib = _lib.lib()
            dx_raw = torch.empty_like(raw)
            dz = torch.empty_like(raw) if (has_res and relu) else None
            dgamma, dbeta = gy.new_empty(cout), gy.new_empty(cout)
            coef = gy.new_empty(cout, 3)
            ws_bytes = lib.az_bn3d_bwd_workspace(nvox, cout)
            ws = gy.new_empty(ws_bytes // 4)
            with profiler.scope(f"bn3d_bwd_{cout}", bytes=4.0 * raw.numel() * (7 if (relu and y is not None) else 5), bound="hbm"):
                _call("az_bn3d_bwd", _p(dx_raw), _p(dz), _p(dgamma), _p(dbeta), _p(coef), _p(ws), ws_bytes,
                      _p(gy), _p(y), _p(raw), _p(mean), _p(invstd), _p(gamma.detach()), _p(scale), _p(shift),
                      int(relu), nvox, cout, _stream())
            g_res = (dz if relu else gy) if has_res else None
            gx = gw = None
            if ctx.needs_input_grad[0]:
                if mode == CONV_S1:    # flipped taps, channels swapped
                    pk = _pack(weight, cout, cin, 27, cin * 27, True)
                    gx = _run_gather(dx_raw, pk, CONV_S1, cout, cin, tag="dgrad")
                elif mode == CONV_S2:  # transposed conv of dy with W[co][ci][k]
                    pk = _pack(weight, cout, cin, 27, cin * 27, False)
                    gx = _run_gather(dx_raw, pk, DECONV_S2, cout, cin, tag="dgrad")
                else:                  # stride-2 conv of dy with Wt[ci][co][k]
                    pk = _pack(weight, cout, cin, cout * 27, 27, False)
                    gx = _run_gather(dx_raw, pk, CONV_S2, cout, cin, tag="dgrad")
            if ctx.needs_input_grad[1]:
                if mode == DECONV_S2:
                    gw = _wgrad(x, dx_raw, 2, cin, cout, "deconv")
                else:
                    gw = _wgrad(dx_raw, x, 1 if mode == CONV_S1 else 2, cout, cin, "conv")
        return gx, gw, dgamma, dbeta, g_res, None, None, None, None


def conv_bn(x, conv, bn, mode, relu=False, residual=None):
    if isinstance(x, LazyCostVolume):  # inference: BN folded, operand synthesised in-kernel
        if torch.is_grad_enabled() or bn.training:
            raise RuntimeError("LazyCostVolume is an inference-only operand")
        with torch.cuda.device(x.fl.device):
            packed, cin, cout = _pack_forward(conv.weight, mode)
            scale, shift = eval_affine(bn, x.fl)
            return _run_gather(x, packed, mode, cin, cout, scale, shift, residual, relu)
    return _ConvBN.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, mode, relu,
                         torch.is_grad_enabled())


class _ConvLogits(torch.autograd.Function):
    """logits = Conv3d(32 -> 1)(x) + addend  (classifN[2] and the running cost sums)."""

    @staticmethod
    def forward(ctx, x, weight, addend):
        x = _chk(x, "x")
        w = _chk(weight.detach().contiguous(), "weight")
        b, d, h, wd, c = _dims(x)
        if c != 32 or tuple(w.shape) != (1, 32, 3, 3, 3):
            raise RuntimeError("classifier conv expects 32 -> 1 channels, 3x3x3")
        if addend is not None:
            addend = _chk(addend.contiguous(), "addend")
        out = x.new_empty(b, d, h, wd)
        with torch.cuda.device(x.device):
            with profiler.scope("conv3d_c1_fwd", bytes=4.0 * (x.numel() + out.numel()), bound="hbm"):
                _call("az_conv3d_c1_fwd", _p(out), _p(x), _p(w), _p(addend), b, d, h, wd, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_add = addend is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _chk(g.contiguous(), "grad_logits")
        b, d, h, wd, _ = _dims(x)
        gx = gw = None
        with torch.cuda.device(g.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                with profiler.scope("conv3d_c1_dgrad", bytes=4.0 * (x.numel() + g.numel()), bound="hbm"):
                    _call("az_conv3d_c1_dgrad", _p(gx), _p(g), _p(w), b, d, h, wd, _stream())
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(w)
                with profiler.scope("conv3d_c1_wgrad", bytes=4.0 * (x.numel() + g.numel()), bound="hbm"):
                    _call("az_conv3d_c1_wgrad", _p(gw), _p(x), _p(g), b, d, h, wd, _stream())
        return gx, gw, (g if ctx.has_add else None)


def conv_logits(x, conv, addend=None):
    return _ConvLogits.apply(x, conv.weight, addend)


class _AddRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "a"), _chk(b, "b")
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _call("az_add_relu", _p(y), _p(a), _p(b), 0, a.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddRelu.apply(a, b)


class _FanOut(torch.autograd.Function):
    """n aliases of x whose gradients are summed in ONE kernel pass (az_sum4) instead of autograd's
    n-1 pairwise adds: for a V0 tensor with four consumers that is 5 tensor passes instead of 9."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [_chk(g.contiguous(), "grad") for g in grads if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            take, gs = gs[:4], gs[4:]
            out = torch.empty_like(take[0])
            ptrs = [_p(t) for t in take] + [None] * (4 - len(take))
            with torch.cuda.device(out.device):
                _call("az_sum4", _p(out), ptrs[0], ptrs[1], ptrs[2], ptrs[3], out.numel(), _stream())
            gs.insert(0, out)
        return gs[0], None


def fanout(x, n):
    """n views of x for n consumers (training); see _FanOut."""
    if not (torch.is_grad_enabled() and x.requires_grad) or x.numel() % 4:
        return (x,) * n
    return _FanOut.apply(x, n)


def conv_plain(x, weight, mode):
    """Bare convolution (no BN), forward only -- used by parity tests and tools."""
    packed, cin, cout = _pack_forward(weight, mode)
    return _run_gather(_chk(x, "x"), packed, mode, cin, cout)


# ----------------------------------------------------------------------------
# 2-D layers of the adjacent feature extractor on the same kernels
# ----------------------------------------------------------------------------
class _Lift2d(torch.autograd.Function):
    """[Cout,Cin,3,3] -> [Cout,Cin,3,3,3] with the 2-D kernel in the centre depth slice.
    A [B,C,H,W] channels-last image is a [B,1,H,W,C] volume; with D = 1 the kd = 0 / kd = 2
    planes are zero padding and the gather / wgrad kernels skip them, so a 3x3 stride-1 conv2d
    costs exactly its own 9 taps."""

    @staticmethod
    def forward(ctx, w2d):
        w3 = w2d.new_zeros(*w2d.shape[:2], 3, 3, 3)
        w3[:, :, 1] = w2d
        return w3

    @staticmethod
    def backward(ctx, g3):
        return g3[:, :, 1].contiguous()


# AZ_FE2D_CH: channel counts of the 2-D layers routed to the 3-D kernels when AZ_FE2D=hip
_FE2D_CH = tuple(int(c) for c in os.environ.get("AZ_FE2D_CH", "64").split(","))


def supports_2d(conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.dilation == (1, 1) and conv.padding == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.in_channels in _FE2D_CH and conv.out_channels in _FE2D_CH)


def _as_volume(x):
    """[B,C,H,W] (any strides) -> contiguous channels-last volume [B,1,H,W,C]; zero-copy when
    the tensor already is torch.channels_last."""
    return x.permute(0, 2, 3, 1).contiguous().unsqueeze(1)


def _as_image(v):
    """[B,1,H,W,C] volume -> [B,C,H,W] tensor in channels_last memory format (a view)."""
    return v.squeeze(1).permute(0, 3, 1, 2)


class _Conv2dS1(torch.autograd.Function):
    """conv2d(x, w) for a 3x3, stride-1, pad-1, dilation-1 layer with 32 or 64 channels on the bf16x6 gather
    kernels (the image is a depth-1 volume).  Backward: input gradient on the same kernels; weight
    gradient on the bf16x6 wgrad kernel or, with AZ_FE2D_WGRAD=miopen, through ATen."""

    @staticmethod
    def forward(ctx, x, w2d):
        xv = _chk(_as_volume(x), "x")
        w3 = w2d.detach().new_zeros(*w2d.shape[:2], 3, 3, 3)
        w3[:, :, 1] = w2d.detach()
        packed, cin, cout = _pack_forward(w3, CONV_S1)
        y = _run_gather(xv, packed, CONV_S1, cin, cout, tag="fe2d")
        ctx.save_for_backward(xv, w3)
        ctx.cfg = (cin, cout)
        return _as_image(y)

    @staticmethod
    def backward(ctx, gy):
        xv, w3 = ctx.saved_tensors
        cin, cout = ctx.cfg
        gv = _chk(_as_volume(gy), "grad_y")
        gx = gw = None
        if ctx.needs_input_grad[0]:
            pk = _pack(w3, cout, cin, 27, cin * 27, True)
            gx = _as_image(_run_gather(gv, pk, CONV_S1, cout, cin, tag="fe2d_dgrad"))
        if ctx.needs_input_grad[1]:
            if os.environ.get("AZ_FE2D_WGRAD", "miopen") == "miopen":
                gw = torch.ops.aten.convolution_backward(
                    gy, _as_image(xv), w3[:, :, 1], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                    [False, True, False])[1]
            else:
                gw = _wgrad(gv, xv, 1, cout, cin, "fe2d")[:, :, 1].contiguous()
        return gx, gw


def conv2d_s1(x, conv):
    """conv(x) for a supports_2d() layer, differentiable; BatchNorm is applied by the caller (bn2d)."""
    return _Conv2dS1.apply(x, conv.weight)


def conv_bn_2d(x, conv, bn, relu=False, residual=None):
    """relu?(BatchNorm2d(Conv2d 3x3 s1 p1 (x)) + residual) on the MFMA gather kernels."""
    if not supports_2d(conv):
        raise RuntimeError("conv_bn_2d: unsupported layer")
    res = _as_volume(residual) if residual is not None else None
    y = _ConvBN.apply(_as_volume(x), _Lift2d.apply(conv.weight), bn.weight, bn.bias, res, bn, CONV_S1,
                      relu, torch.is_grad_enabled())
    return _as_image(y)
