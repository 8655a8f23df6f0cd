#!/usr/bin/env python3
"""The "PyTorch-eager on MI355X" denominator of BASELINE.md (north_star: ">= 5x the PyTorch-eager
cost-volume + 3-D conv throughput"): the reference's forward (nets/psmnet/psmnet_3.py:144-220) written
with stock PyTorch-ROCm operators (MIOpen convolutions, ATen batch_norm / interpolate / softmax) over
the PRODUCT module's own parameters, so that the two paths share weights and data.  Measurement
tooling: lives outside activezero_amd/ on purpose (the product has no vendor-library backend)."""
import torch
import torch.nn.functional as F


def _cost_volume(fl, fr, nd):  # psmnet_3.py:149-163 (allocated on the device: no 1.6 GB host copy)
    b, c, h, w = fl.shape
    vol = fl.new_zeros(b, 2 * c, nd, h, w)
    for i in range(nd):
        if i > 0:
            vol[:, :c, i, :, i:] = fl[:, :, :, i:]
            vol[:, c:, i, :, i:] = fr[:, :, :, :-i]
        else:
            vol[:, :c, i] = fl
            vol[:, c:, i] = fr
    return vol.contiguous()


def _fe(fe, x):  # psmnet_submodule_3.py:186-220 with the module tree's own (standard) nn layers
    seq = lambda mods, t: torch.nn.Sequential(*mods)(t)
    y = fe.firstconv(x)

    def block(blk, t):
        out = blk.conv2(blk.conv1(t))
        return out + (t if blk.downsample is None else blk.downsample(t))

    for layer in (fe.layer1, fe.layer2):
        for blk in layer:
            y = block(blk, y)
    raw = y
    for layer in (fe.layer3, fe.layer4):
        for blk in layer:
            y = block(blk, y)
    skip = y
    size = skip.shape[-2:]
    pyr = [F.interpolate(getattr(fe, f"branch{i}")(skip), size, mode="bilinear", align_corners=True)
           for i in (4, 3, 2, 1)]
    return fe.lastconv(torch.cat([raw, skip] + pyr, 1))


def _hourglass(hg, x, presqu, postqu):  # psmnet_3.py:36-58
    out = hg.conv1(x)
    pre = hg.conv2(out)
    pre = F.relu(pre + postqu if postqu is not None else pre, inplace=True)
    out = hg.conv4(hg.conv3(pre))
    post = F.relu(hg.conv5(out) + (presqu if presqu is not None else pre), inplace=True)
    return hg.conv6(post), pre, post


def _head(cost, maxdisp, h, w):  # psmnet_3.py:184-215 + psmnet_submodule_3.py:80-89
    up = F.interpolate(cost, (maxdisp, h, w), mode="trilinear", align_corners=False)
    prob = F.softmax(up.squeeze(1), dim=1)
    ramp = torch.arange(maxdisp, dtype=prob.dtype, device=prob.device).view(1, -1, 1, 1)
    return torch.sum(prob * ramp, 1, keepdim=True)


def eager_forward(model, img_l, img_r):
    """model: activezero_amd.nets.psmnet.psmnet_3.PSMNet (its parameters / buffers are used and updated)."""
    fl, fr = _fe(model.feature_extraction, img_l), _fe(model.feature_extraction, img_r)
    return eager_from_features(model, fl, fr, img_l.shape[-2:])


def eager_from_features(model, fl, fr, size):
    """psmnet_3.py:149-220: cost volume + 3-D aggregation + soft-argmin heads from the two feature maps"""
    cost = _cost_volume(fl, fr, model.maxdisp // 4)
    cost0 = model.dres0(cost)
    cost0 = model.dres1(cost0) + cost0
    out1, pre1, post1 = _hourglass(model.dres2, cost0, None, None)
    out1 = out1 + cost0
    out2, _, post2 = _hourglass(model.dres3, out1, pre1, post1)
    out2 = out2 + cost0
    out3, _, _ = _hourglass(model.dres4, out2, pre1, post2)
    out3 = out3 + cost0
    cost1 = model.classif1(out1)
    cost2 = model.classif2(out2) + cost1
    cost3 = model.classif3(out3) + cost2
    h, w = size
    pred3 = _head(cost3, model.maxdisp, h, w)
    if model.training:
        return pred3, _head(cost2, model.maxdisp, h, w), _head(cost1, model.maxdisp, h, w)
    return pred3


def eager_loss(preds, gt, maxdisp):  # utils/losses.py:7-15 with the train.py:272 mask, eager ops
    mask = (gt < maxdisp) & (gt > 0)
    sl1 = lambda p: F.smooth_l1_loss(p[mask], gt[mask], reduction="mean")
    p3, p2, p1 = preds
    return 0.5 * sl1(p1) + 0.7 * sl1(p2) + sl1(p3)
