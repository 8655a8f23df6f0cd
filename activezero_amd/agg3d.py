"""3-D cost aggregation executor.

PSMNet's hourglass / dres / classif blocks (reference nets/psmnet/psmnet_3.py:11-77,
87-117, 165-179) are expressed on top of five primitives working on channels-last
volumes [B,D,H,W,C]:

    costvol_conv_bn(feat_l, feat_r, ndisp, unit)  K3' concat cost volume o dres0[0], never materialised
    conv_bn(vol, unit, relu, add)                 K4  Conv3d + BatchNorm3d (+residual, +ReLU)
    deconv_bn(vol, unit, relu, add)               K5  ConvTranspose3d + BatchNorm3d (...)
    conv_logits(vol, conv, add)                   K4  32 -> 1 classifier (+ running cost sum)
    add(a, b) / fanout(x, n)                      residual sum / fused gradient sum of n consumers

`unit` is the nn.Sequential(Conv3d|ConvTranspose3d, BatchNorm3d) parameter container
whose names match the reference state-dict; the modules' own forward() is never
called.  Every function is a pure function of its arguments (`arith` selects the MFMA
arithmetic, conv3d.Arith): there is no backend switch and no process-wide state -- the
PyTorch-eager denominator of BASELINE.md lives in tools/eager_psmnet.py, outside the product.
Logits are returned as [B, d, h, w].
"""
import torch

from . import bn2d, conv3d, costconv, ops


def _mode_of(conv):
    if isinstance(conv, torch.nn.ConvTranspose3d):
        return conv3d.DECONV_S2
    return conv3d.CONV_S1 if conv.stride[0] == 1 else conv3d.CONV_S2


def volume_from_features(feat_l, feat_r, ndisp, lazy=None):
    """The concat cost volume (psmnet_3.py:149-163) for callers that want dres0[0] as a 64 -> 32
    3-D convolution (PSMNet itself uses costvol_conv_bn).  lazy=True (default when autograd is
    off): a recipe the convolution kernel expands on the fly; else the NDHWC tensor (K3)."""
    fl = feat_l.permute(0, 2, 3, 1).contiguous()
    fr = feat_r.permute(0, 2, 3, 1).contiguous()
    if lazy is None:
        lazy = not torch.is_grad_enabled()
    if lazy:
        return conv3d.LazyCostVolume(fl, fr, ndisp)
    return ops.cost_volume_ndhwc(fl, fr, ndisp)


def costvol_conv_bn(feat_l, feat_r, ndisp, unit, relu=False, arith=None):
    """relu?(BatchNorm3d(Conv3d(64,32,3,pad 1)(concat cost volume))) straight from the two [B,32,h,w]
    feature maps (psmnet_3.py:149-166): returns the [B,ndisp,h,w,32] activation, no volume in between."""
    conv, bn = unit[0], unit[1]
    raw = costconv.costvol_conv(feat_l, feat_r, ndisp, conv.weight, arith)  # [B,D,h,w,32]
    b, d, h, w, c = raw.shape
    y = bn2d.bn_act(raw.view(b, d * h, w, c).permute(0, 3, 1, 2), bn, relu=relu)
    return y.permute(0, 2, 3, 1).reshape(b, d, h, w, c)


def conv_bn(vol, unit, relu=False, add=None, arith=None, defer=None):
    return conv3d.conv_bn(vol, unit[0], unit[1], _mode_of(unit[0]), relu, add, arith, defer)


def deconv_bn(vol, unit, relu=False, add=None, arith=None):
    return conv3d.conv_bn(vol, unit[0], unit[1], conv3d.DECONV_S2, relu, add, arith)


def conv_logits(vol, conv, add=None, arith=None, affine=None):
    return conv3d.conv_logits(vol, conv, add, arith.sink if arith is not None else None, affine)


def add(a, b):
    return conv3d.add(a, b)


def fanout(x, n):
    """n handles on x for n consumers; their gradients are summed in one pass."""
    return conv3d.fanout(x, n)
