"""2-D feature extractor + building blocks of PSMNet (3-channel input).

Mirrors the public names of the reference module nets/psmnet/psmnet_submodule_3.py (convbn, conv,
convbn_3d, BasicBlock, DisparityRegression, FeatureExtraction) and its parameter/buffer names, so
reference checkpoints load unchanged.  The modules are parameter containers: forward() runs every
convolution on the hand-written 2-D MFMA kernels (activezero_amd/conv2d.py) and every
BatchNorm (+ReLU, +residual) on the HIP BatchNorm kernels (activezero_amd/bn2d.py); there is no
vendor-library convolution and no fallback.  The 3-D blocks built from convbn_3d are executed by
activezero_amd.agg3d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from activezero_amd import bn2d, conv2d


def _convbn_unit(x, unit, relu=False, residual=None, groups=1, skip=False, arith=None):
    """unit = Sequential(Conv2d, BatchNorm2d): y = relu?(bn(conv(x)) + residual).
    skip=True returns (y, x'), x' = x routed through the convolution's autograd node for the caller's
    shortcut, so that the shortcut's gradient is added inside the input-gradient kernel (conv2d._ConvSame).
    `arith`: conv3d.Arith of the pass (arithmetic of the stride-2 route, weight-gradient sink), None = default.
    `groups`: consecutive equal parts of the batch that take their OWN batch statistics (2 when the left
    and right images run as one stacked batch, FeatureExtraction.forward_pair); passed down explicitly --
    no module-level state, so replicas on several threads (nn.DataParallel, train.py:540-541) cannot
    disturb each other."""
    conv, bn = unit[0], unit[1]
    if not x.is_cuda:
        raise RuntimeError("the feature extractor runs on the GPU only (no CPU fallback)")
    if not bn2d.supported(bn, x):
        raise RuntimeError(f"unsupported BatchNorm2d for the HIP path: {bn}")
    training = bn.training or not bn.track_running_stats
    if not training and not torch.is_grad_enabled():
        # inference: BatchNorm (running statistics), residual sum and ReLU ride on the conv epilogue
        y = conv2d.conv_bn_eval(x, conv, bn, relu, residual)
        if y is not None:
            return (y, x) if skip else y
    # train mode: the convolution's epilogue also reduces the BatchNorm partials of its output
    stats = bn2d.Partials(groups) if training else None
    if skip and torch.is_grad_enabled() and x.requires_grad and conv2d.is_same(conv):
        c, shortcut = conv2d.conv(x, conv, arith, skip=True, stats=stats)
        return bn2d.bn_act(c, bn, relu, residual, groups, stats), shortcut
    y = bn2d.bn_act(conv2d.conv(x, conv, arith, stats=stats), bn, relu, residual, groups, stats)
    return (y, x) if skip else y


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

    def forward(self, x, groups=1, arith=None):
        if self.downsample is None:
            y, shortcut = _convbn_unit(x, self.conv1[0], relu=True, groups=groups, skip=True, arith=arith)
        else:
            shortcut = _convbn_unit(x, self.downsample, groups=groups, arith=arith)
            y = _convbn_unit(x, self.conv1[0], relu=True, groups=groups, arith=arith)
        return _convbn_unit(y, self.conv2, relu=False, residual=shortcut, groups=groups, arith=arith)


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


class _SppConcat(torch.autograd.Function):
    """torch.cat([raw, skip] + [F.upsample(branch_i, (H, W), mode="bilinear")], 1) of psmnet_submodule_3.py:198-211 as one node:
    the four pooled branch maps are interpolated straight into their channel slots of the concat buffer by az_spp_upsample_fwd
    (a 2 x 2-tap stencil; rounds 1-4: two rocBLAS GEMMs and a layout copy per branch, forward and backward), raw and skip are
    copied into theirs.  Channels-last throughout: out is [B, 64 + 128 + 4 * 32, H, W] in channels_last memory."""

    @staticmethod
    def forward(ctx, raw, skip, *branches):
        from activezero_amd.ops import _call, _chk, _p, _stream
        b, _, h, w = skip.shape
        parts = [raw, skip]
        ctot = raw.shape[1] + skip.shape[1] + sum(t.shape[1] for t in branches)
        out = skip.new_empty(b, h, w, ctot)  # rows
        at = 0
        for t in parts:
            out[..., at:at + t.shape[1]] = t.permute(0, 2, 3, 1)
            at += t.shape[1]
        ctx.slots, ctx.shapes = [], []
        with torch.cuda.device(skip.device):
            for t in branches:
                r = _chk(conv2d.rows(t), "spp branch")
                c = r.shape[-1]
                _call("az_spp_upsample_fwd", _p(out[..., at:]), _p(r), b, r.shape[1], r.shape[2], h, w, c, ctot, _stream())
                ctx.slots.append((at, c))
                ctx.shapes.append(tuple(r.shape))
                at += c
        ctx.c_raw, ctx.c_skip = raw.shape[1], skip.shape[1]
        return conv2d.image(out)

    @staticmethod
    def backward(ctx, g):
        from activezero_amd import _lib
        from activezero_amd.ops import _call, _chk, _p, _stream
        gr = _chk(conv2d.rows(g), "grad")
        b, h, w, ctot = gr.shape
        need = ctx.needs_input_grad
        g_raw = conv2d.image(gr[..., :ctx.c_raw]) if need[0] else None
        g_skip = conv2d.image(gr[..., ctx.c_raw:ctx.c_raw + ctx.c_skip]) if need[1] else None
        outs = []
        with torch.cuda.device(g.device):
            for k, ((at, c), shp) in enumerate(zip(ctx.slots, ctx.shapes)):
                if not need[2 + k]:
                    outs.append(None)
                    continue
                gi = gr.new_empty(shp)
                wsb = _lib.lib().az_spp_upsample_bwd_workspace(b, shp[2], h, c)
                wsp = gr.new_empty(wsb // 4)
                _call("az_spp_upsample_bwd", _p(gi), _p(wsp), wsb, _p(gr[..., at:]), b, shp[1], shp[2], h, w, c, ctot, _stream())
                outs.append(conv2d.image(gi))
        return (g_raw, g_skip, *outs)


def spp_concat(raw, skip, branches):
    """[raw | skip | upsampled branches] along the channels (see _SppConcat); branches: [B,32,hs,ws] maps, any size"""
    return _SppConcat.apply(raw, skip, *branches)


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

    def _trunk(self, x, groups=1, arith=None):
        g, a = groups, arith
        y = _convbn_unit(x, self.firstconv[0], relu=True, groups=g, arith=a)
        y = _convbn_unit(y, self.firstconv[2], relu=True, groups=g, arith=a)
        y = _convbn_unit(y, self.firstconv[4], relu=True, groups=g, arith=a)
        for blk in self.layer1:
            y = blk(y, g, a)
        for blk in self.layer2:
            y = blk(y, g, a)
        raw = y
        for blk in self.layer3:
            y = blk(y, g, a)
        for blk in self.layer4:
            y = blk(y, g, a)
        skip = y
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
        branches = [_convbn_unit(pooled[win_of[i]].contiguous(memory_format=torch.channels_last),
                                 getattr(self, f"branch{i}")[1], relu=True, groups=g, arith=a) for i in (4, 3, 2, 1)]
        # upsampling + concat in one node, the branches written straight into their slots (_SppConcat)
        y = _convbn_unit(spp_concat(raw, skip, branches), self.lastconv[0], relu=True, groups=g, arith=a)
        return conv2d.conv(y, self.lastconv[2], a)

    def forward(self, x, arith=None):
        """[B,3,H,W] -> [B,32,H/4,W/4]"""
        return self._trunk(x.contiguous(memory_format=torch.channels_last), arith=arith)

    def forward_pair(self, left, right, arith=None):
        """(feature_extraction(left), feature_extraction(right)) of psmnet_3.py:145-146 in ONE pass
        over the stacked batch: every BatchNorm takes its statistics per image set and updates its
        running statistics left first, then right, exactly as the two sequential calls do."""
        if left.shape != right.shape:
            return self.forward(left, arith), self.forward(right, arith)
        x = torch.cat([left, right], 0).contiguous(memory_format=torch.channels_last)
        y = self._trunk(x, groups=2, arith=arith)
        b = left.shape[0]
        return y[:b], y[b:]
