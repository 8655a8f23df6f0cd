"""3-D cost aggregation executor.

PSMNet's hourglass / dres / classif blocks (reference nets/psmnet/psmnet_3.py:11-77,
87-117, 165-179) are expressed on top of five primitives working on an opaque
"volume" handle:

    volume_from_features(feat_l, feat_r, ndisp)   K3  concat cost volume
    conv_bn(vol, unit, relu, add)                 K4  Conv3d + BatchNorm3d (+residual, +ReLU)
    deconv_bn(vol, unit, relu, add)               K5  ConvTranspose3d + BatchNorm3d (...)
    conv_logits(vol, conv, add)                   K4  32 -> 1 classifier (+ running cost sum)
    add(a, b)                                     plain residual sum

`unit` is the nn.Sequential(Conv3d|ConvTranspose3d, BatchNorm3d) parameter container
whose names match the reference state-dict; the modules' own forward() is never
called.

Backends (env AZ_AGG3D):
  hip     (default) hand-written gfx950 kernels, volumes are channels-last [B,D,H,W,C]
  miopen  PyTorch-ROCm / MIOpen operators on NCDHW tensors: the "PyTorch-eager"
          baseline of BASELINE.md kept for A/B measurements (bench.py --backend miopen).
          It is a GPU library path, not a CPU fallback.
Both return logits as [B, d, h, w].
"""
import os

import torch
import torch.nn.functional as F

from . import bn2d, conv3d, costconv, ops, profiler

BACKEND = os.environ.get("AZ_AGG3D", "hip")
FUSE_COST_VOLUME = os.environ.get("AZ_FUSE_COSTVOL", "1") != "0"


def set_backend(name):
    global BACKEND
    if name not in ("hip", "miopen"):
        raise ValueError(name)
    BACKEND = name


# ------------------------------------------------------------------ hip backend
def _mode_of(conv):
    if isinstance(conv, torch.nn.ConvTranspose3d):
        return conv3d.DECONV_S2
    return conv3d.CONV_S1 if conv.stride[0] == 1 else conv3d.CONV_S2


def volume_from_features(feat_l, feat_r, ndisp):
    if BACKEND == "miopen":
        return ops.cost_volume(feat_l, feat_r, ndisp)
    # features arrive NCHW from the 2-D extractor; the 3-D kernels want channels-last
    fl = feat_l.permute(0, 2, 3, 1).contiguous()
    fr = feat_r.permute(0, 2, 3, 1).contiguous()
    if not torch.is_grad_enabled() and FUSE_COST_VOLUME:
        # inference: never materialise the 401 MB/pair volume (reference psmnet_3.py:149-163);
        # dres0[0] builds its operand from the two feature maps in-kernel
        return conv3d.LazyCostVolume(fl, fr, ndisp)
    return ops.cost_volume_ndhwc(fl, fr, ndisp)


# AZ_COSTCONV=0: materialise the cost volume (training) / synthesise it inside the gather kernel
# (inference) and run dres0[0] as a 64 -> 32 3-D convolution; default: the factored form (costconv.py)
FACTORED_COSTCONV = os.environ.get("AZ_COSTCONV", "1") != "0"


def use_costconv(feat_l):
    return BACKEND == "hip" and FACTORED_COSTCONV and feat_l.is_cuda and feat_l.shape[1] == 32


def costvol_conv_bn(feat_l, feat_r, ndisp, unit, relu=False):
    """relu?(BatchNorm3d(Conv3d(64,32,3,pad 1)(concat cost volume))) straight from the two [B,32,h,w]
    feature maps (psmnet_3.py:149-166): returns the [B,ndisp,h,w,32] activation, no volume in between."""
    conv, bn = unit[0], unit[1]
    raw = costconv.costvol_conv(feat_l, feat_r, ndisp, conv.weight)  # [B,D,h,w,32]
    b, d, h, w, c = raw.shape
    y = bn2d.bn_act(raw.view(b, d * h, w, c).permute(0, 3, 1, 2), bn, relu=relu)
    return y.permute(0, 2, 3, 1).reshape(b, d, h, w, c)


def conv_bn(vol, unit, relu=False, add=None):
    if BACKEND == "miopen":
        return _miopen_conv_bn(vol, unit, relu, add)
    return conv3d.conv_bn(vol, unit[0], unit[1], _mode_of(unit[0]), relu, add)


def deconv_bn(vol, unit, relu=False, add=None):
    if BACKEND == "miopen":
        return _miopen_deconv_bn(vol, unit, relu, add)
    return conv3d.conv_bn(vol, unit[0], unit[1], conv3d.DECONV_S2, relu, add)


def conv_logits(vol, conv, add=None):
    if BACKEND == "miopen":
        y = F.conv3d(vol, conv.weight, None, conv.stride, conv.padding)[:, 0]
        return y if add is None else y + add
    return conv3d.conv_logits(vol, conv, add)


def add(a, b):
    if BACKEND == "miopen":
        return a + b
    return conv3d.add(a, b)


def fanout(x, n):
    """n handles on x for n consumers; on the HIP backend their gradients are summed in one pass."""
    if BACKEND == "miopen":
        return (x,) * n
    return conv3d.fanout(x, n)


# ------------------------------------------------------------------ miopen backend
def _bn(x, bn):
    training = bn.training or not bn.track_running_stats
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training,
                        bn.momentum, bn.eps)


def _miopen_conv_bn(vol, unit, relu, add):
    cv, bn = unit[0], unit[1]
    with profiler.scope(f"miopen_conv3d_fwd_{cv.in_channels}_{cv.out_channels}_s{cv.stride[0]}",
                        flops=2.0 * 27 * cv.in_channels * cv.out_channels * vol.shape[0]
                        * (vol[0, 0].numel() // cv.stride[0] ** 3)):
        y = F.conv3d(vol, cv.weight, None, cv.stride, cv.padding)
    y = _bn(y, bn)
    if add is not None:
        y = y + add
    return F.relu(y) if relu else y


def _miopen_deconv_bn(vol, unit, relu, add):
    dc, bn = unit[0], unit[1]
    y = F.conv_transpose3d(vol, dc.weight, None, dc.stride, dc.padding, dc.output_padding)
    y = _bn(y, bn)
    if add is not None:
        y = y + add
    return F.relu(y) if relu else y
