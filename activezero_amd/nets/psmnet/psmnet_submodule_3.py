"""2-D feature extractor + building blocks of PSMNet (3-channel input).

Mirrors the public names of the reference module
nets/psmnet/psmnet_submodule_3.py (convbn, conv, convbn_3d, BasicBlock,
DisparityRegression, FeatureExtraction) and its parameter/buffer names, so
reference checkpoints load unchanged.  The 2-D ResNet+SPP extractor is the
"adjacent" stage of SURVEY.md 8f: ordinary PyTorch-ROCm (MIOpen) modules.
The 3-D blocks built from convbn_3d are executed by activezero_amd.agg3d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from activezero_amd import bn2d, conv3d, ops

# 2-D stage backend.  "miopen" (default): PyTorch-ROCm/MIOpen modules, channels-last.
# "hip" (opt-in experiment): the 3x3 stride-1 32/64-channel conv+BN(+ReLU)(+residual) units of
# the extractor run on the MFMA gather kernels as D=1 volumes (conv3d.conv_bn_2d).  Measured on
# MI355X at B=4, 544x960: 190.2 ms/step vs 184.9 ms with MIOpen -- the 2-D layers are small
# (<= 10 GFLOP each) and launch/occupancy bound on the 3-D tiling, so MIOpen stays the default
# until the 2-D stage gets its own tiling.
# AZ_FE2D: "fused" (default) = MIOpen convolutions + the HIP BatchNorm kernels with ReLU / residual
# folded into the normalisation pass and per-group batch statistics (left and right images run as
# ONE batch of 2B, activezero_amd/bn2d.py); "miopen" = plain torch modules, two passes; "hip" = the
# stride-1 3x3 layers on the 3-D gather kernels (experiment, slower).
FE2D_BACKEND = os.environ.get("AZ_FE2D", "fused")
FE2D_CONV = os.environ.get("AZ_FE2D_CONV", "hip")
# inference experiments: "fold" = BatchNorm folded into the conv weights + bias (plain conv2d, cached);
# "fused" = the same through MIOpen's conv+bias+ReLU fusion; "" (default) = the HIP BatchNorm apply pass.
# Measured (eval forward): 256x512/D=64 4.35 ms -> 3.54 (fold) / 155 (fused); 540x960/D=192 12.7 ms ->
# 13.3 (fold) / 540 (fused): the fused apply(+ReLU+residual) pass wins at the sizes that matter.
FE2D_EVAL_FOLD = os.environ.get("AZ_FE2D_EVAL_FOLD", "")
_STAT_GROUPS = 1  # batch-statistic groups of the pass in flight (2 inside forward_pair)


_FOLD_CACHE = {}


def _folded(conv, bn):
    """(weight * scale[c_out], shift) of an eval-mode conv+BN unit, cached on the parameters' versions."""
    ts = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in ts)
    hit = _FOLD_CACHE.get(key)
    if hit is None:
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w_f = (conv.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
        b_f = bn.bias - bn.running_mean * scale
        if len(_FOLD_CACHE) > 256:
            _FOLD_CACHE.clear()
        hit = _FOLD_CACHE[key] = (w_f, b_f) + ts  # (holds its sources: their addresses stay unique)
    return hit[0], hit[1]


def _convbn_unit(x, unit, relu=False, residual=None):
    """unit = Sequential(Conv2d, BatchNorm2d): y = relu?(bn(conv(x)) + residual)"""
    conv, bn = unit[0], unit[1]
    if FE2D_BACKEND == "hip" and x.is_cuda and conv3d.supports_2d(conv) and _STAT_GROUPS == 1:
        return conv3d.conv_bn_2d(x, conv, bn, relu, residual)
    if (FE2D_EVAL_FOLD and FE2D_BACKEND != "miopen" and x.is_cuda and not bn.training
            and not torch.is_grad_enabled() and bn.track_running_stats):
        # inference: BatchNorm folded into the convolution's weights and bias (cached), ReLU fused by
        # MIOpen where the layer has one -- no separate normalisation pass
        w_f, b_f = _folded(conv, bn)
        if relu and residual is None and FE2D_EVAL_FOLD == "fused":
            return torch.ops.aten.miopen_convolution_relu(x, w_f, b_f, conv.stride, conv.padding, conv.dilation, 1)
        y = F.conv2d(x, w_f, b_f, conv.stride, conv.padding, conv.dilation)
        if residual is not None:
            y = y + residual
        return F.relu_(y) if relu else y
    if FE2D_BACKEND != "miopen" and bn2d.supported(bn, x):
        # AZ_FE2D_CONV=hip (default): the stride-1 3x3 64-channel layers (layer2, a third of the extractor's
        # FLOPs) run forward and input gradient on the bf16x6 gather kernel as depth-1 volumes; their weight
        # gradient stays on MIOpen (AZ_FE2D_WGRAD).  AZ_FE2D_CONV=miopen: every convolution on MIOpen.
        y = conv3d.conv2d_s1(x, conv) if (FE2D_CONV == "hip" and conv3d.supports_2d(conv)) else conv(x)
        return bn2d.bn_act(y, bn, relu, residual, _STAT_GROUPS)
    if _STAT_GROUPS != 1:
        raise RuntimeError("grouped batch statistics need the fused BatchNorm path")
    y = unit(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y

__all__ = ["convbn", "conv", "convbn_3d", "BasicBlock", "DisparityRegression",
           "FeatureExtraction", "torch", "nn", "F"]


def _conv2d(cin, cout, k, stride, pad, dilation):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, dilation=dilation,
                     padding=dilation if dilation > 1 else pad, bias=False)


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
    return nn.Sequential(_conv2d(in_planes, out_planes, kernel_size, stride, pad, dilation),
                         nn.BatchNorm2d(out_planes))


def conv(in_planes, out_planes, kernel_size, stride, pad, dilation):
    return nn.Sequential(_conv2d(in_planes, out_planes, kernel_size, stride, pad, dilation))


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    """Parameter container (Conv3d weight + BatchNorm3d affine/running stats) for one
    3-D conv+BN unit; reference psmnet_submodule_3.py:44-56."""
    return nn.Sequential(
        nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=pad,
                  bias=False),
        nn.BatchNorm3d(out_planes))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation),
                                   nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else _convbn_unit(x, self.downsample)
        y = _convbn_unit(x, self.conv1[0], relu=True)
        return _convbn_unit(y, self.conv2, relu=False, residual=shortcut)


class DisparityRegression(nn.Module):
    """sum_d d * p[b,d,y,x] (reference psmnet_submodule_3.py:80-89).  Kept for API
    parity; PSMNet.forward uses the fused soft-argmin kernel instead of
    softmax + this module."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        ramp = torch.arange(self.maxdisp, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        return torch.sum(x * ramp, 1, keepdim=True)


_SPP_WINDOWS = ((1, 64), (2, 32), (3, 16), (4, 8))
_INTERP_CACHE = {}


def _interp_matrix(n_in, n_out, device):
    """[n_out, n_in] matrix of 1-D linear interpolation with align_corners=True."""
    key = (n_in, n_out, str(device))
    m = _INTERP_CACHE.get(key)
    if m is None:
        # fp32 source positions exactly as ATen computes them: scale = (in-1)/(out-1), src = scale*dst
        scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)
        pos = scale * torch.arange(n_out, dtype=torch.float32)
        i0 = pos.floor().clamp(0, n_in - 1).long()
        i1 = (i0 + 1).clamp(max=n_in - 1)
        frac = pos - i0.to(torch.float32)
        m = torch.zeros(n_out, n_in, dtype=torch.float32)
        rows = torch.arange(n_out)
        m[rows, i0] += 1.0 - frac
        m[rows, i1] += frac
        m = _INTERP_CACHE[key] = m.to(torch.float32).to(device)
    return m


def upsample_bilinear_ac(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=True) for the tiny SPP maps,
    written as two dense products out = Wy @ x @ Wx^T.  Same linear map; its backward is
    two GEMMs as well, instead of ATen's atomics kernel (2.5 ms per call at 136x240)."""
    wy = _interp_matrix(x.shape[-2], size[0], x.device)
    wx = _interp_matrix(x.shape[-1], size[1], x.device)
    # channels_last like every other operand of the torch.cat that follows: a single NCHW input
    # makes cat emit NCHW and the 320-channel tensor is then re-laid-out for lastconv (4 x 0.32 ms
    # per step, forward and backward)
    return torch.matmul(wy, torch.matmul(x, wx.t())).contiguous(memory_format=torch.channels_last)


class FeatureExtraction(nn.Module):
    IN_CHANNELS = 3

    def __init__(self):
        super().__init__()
        act = lambda: nn.ReLU(inplace=True)
        self.inplanes = 32
        self.firstconv = nn.Sequential(
            convbn(self.IN_CHANNELS, 32, 3, 2, 1, 1), act(),
            convbn(32, 32, 3, 1, 1, 1), act(),
            convbn(32, 32, 3, 1, 1, 1), act())
        self.layer1 = self._make_layer(BasicBlock, 32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(BasicBlock, 64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(BasicBlock, 128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(BasicBlock, 128, 3, 1, 1, 2)
        for idx, win in _SPP_WINDOWS:
            setattr(self, f"branch{idx}", nn.Sequential(
                nn.AvgPool2d((win, win), stride=(win, win)), convbn(128, 32, 1, 1, 0, 1), act()))
        self.lastconv = nn.Sequential(
            convbn(320, 128, 3, 1, 1, 1), act(),
            nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))

    def _make_layer(self, block, planes, blocks, stride, pad, dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                          bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, pad, dilation)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _trunk(self, x):
        y = _convbn_unit(x, self.firstconv[0], relu=True)
        y = _convbn_unit(y, self.firstconv[2], relu=True)
        y = _convbn_unit(y, self.firstconv[4], relu=True)
        raw = self.layer2(self.layer1(y))
        skip = self.layer4(self.layer3(raw))
        size = skip.shape[-2:]
        # SPP pooling as a hierarchy: the 16/32/64 windows are 2x2 means of the previous level
        # (floor division composes, so sizes and covered pixels equal AvgPool2d(win, win)); the
        # full-resolution map is read once instead of four times, and written once in backward
        pooled, p = {}, skip
        for _, win in sorted(_SPP_WINDOWS, key=lambda iw: iw[1]):
            p = F.avg_pool2d(p, win, win) if not pooled else F.avg_pool2d(p, 2, 2)
            pooled[win] = p
        assert sorted(pooled) == [8, 16, 32, 64]
        win_of = dict(_SPP_WINDOWS)
        pyramid = [upsample_bilinear_ac(_convbn_unit(pooled[win_of[i]], getattr(self, f"branch{i}")[1], relu=True),
                                        size) for i in (4, 3, 2, 1)]
        y = _convbn_unit(torch.cat([raw, skip] + pyramid, 1), self.lastconv[0], relu=True)
        return self.lastconv[2](y)

    def forward(self, x):
        """[B,3,H,W] -> [B,32,H/4,W/4]"""
        return self._trunk(x)

    def forward_pair(self, left, right):
        """(feature_extraction(left), feature_extraction(right)) of psmnet_3.py:145-146 in ONE pass
        over the stacked batch: every BatchNorm takes its statistics per image set and updates its
        running statistics left first, then right, exactly as the two sequential calls do."""
        global _STAT_GROUPS
        if FE2D_BACKEND == "miopen" or not left.is_cuda or left.shape != right.shape:
            return self._trunk(left), self._trunk(right)
        x = torch.cat([left, right], 0).contiguous(memory_format=torch.channels_last)
        _STAT_GROUPS = 2
        try:
            y = self._trunk(x)
        finally:
            _STAT_GROUPS = 1
        b = left.shape[0]
        return y[:b], y[b:]
