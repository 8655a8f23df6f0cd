"""3-D cost aggregation executor.

PSMNet's hourglass / dres / classif blocks (reference nets/psmnet/psmnet_3.py:11-77,
87-117, 165-179) are expressed on top of four primitives working on an opaque
"volume" handle:

    volume_from_features(feat_l, feat_r, ndisp)        K3
    conv_bn(vol, unit, relu, add)                      K4 (+BN, +ReLU, +residual)
    deconv_bn(vol, unit, relu, add)                    K5
    conv_logits(vol, conv)                             K4 (32 -> 1 classifier)

`unit` is the nn.Sequential(Conv3d|ConvTranspose3d, BatchNorm3d) parameter
container whose names match the reference state-dict.

BACKEND STATUS (round 1): the cost volume (K3) and the soft-argmin head (K6)
run on the hand-written HIP kernels; conv/deconv/BN of K4/K5 are dispatched to
`hip` (activezero_amd.conv3d, hand-written MFMA kernels) when that backend is
enabled and to PyTorch-ROCm's MIOpen operators otherwise.  MIOpen here is the
"PyTorch-eager" baseline of BASELINE.md, not a CPU fallback; DESIGN.md tracks
which layers are native.
"""
import os

import torch
import torch.nn.functional as F

from . import ops

BACKEND = os.environ.get("AZ_AGG3D", "miopen")


def volume_from_features(feat_l, feat_r, ndisp):
    return ops.cost_volume(feat_l, feat_r, ndisp)


def _bn(x, bn, training):
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                        training or not bn.track_running_stats, bn.momentum, bn.eps)


def conv_bn(vol, unit, relu=False, add=None):
    cv, bn = unit[0], unit[1]
    y = F.conv3d(vol, cv.weight, None, cv.stride, cv.padding)
    y = _bn(y, bn, bn.training)
    if add is not None:
        y = y + add
    return F.relu(y) if relu else y


def deconv_bn(vol, unit, relu=False, add=None):
    dc, bn = unit[0], unit[1]
    y = F.conv_transpose3d(vol, dc.weight, None, dc.stride, dc.padding, dc.output_padding)
    y = _bn(y, bn, bn.training)
    if add is not None:
        y = y + add
    return F.relu(y) if relu else y


def conv_logits(vol, conv):
    return F.conv3d(vol, conv.weight, None, conv.stride, conv.padding)


def add(a, b):
    return a + b
