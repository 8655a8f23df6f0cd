"""ORACLE (test infrastructure, never the product path).

CPU restatement, in plain eager PyTorch fp32, of the reference's PSMNet hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file.  Parity pin: tests/golden/*.npz, produced by
tools/make_goldens.py from the reference imported in the build container
(see DESIGN.md "Oracle").

Reference lines restated here (all relative to /root/reference):
  * cost volume ............ nets/psmnet/psmnet_3.py:149-163 (psmnet.py:151-165)
  * dres0/dres1/hourglass .. nets/psmnet/psmnet_3.py:11-77, 87-117, 165-179
  * upsample+softmax ....... nets/psmnet/psmnet_3.py:184-212
  * DisparityRegression .... nets/psmnet/psmnet_submodule_3.py:80-89
  * FeatureExtraction ...... nets/psmnet/psmnet_submodule_3.py:92-220 (adjacent)
  * weight init ............ nets/psmnet/psmnet_3.py:123-142
  * psmnet_disp loss ....... utils/losses.py:7-15 ; mask rule train.py:272

The module tree is table-driven but reproduces the reference's parameter and
buffer names exactly (514 state-dict keys) so the same procedurally generated
weights load into the reference, this oracle and the product module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# functional pieces of the hot path
# --------------------------------------------------------------------------
def build_cost_volume(feat_l, feat_r, ndisp):
    """cost[b, :C, i, y, x] = L[b, :, y, x], cost[b, C:, i, y, x] = R[b, :, y, x - i]
    for x >= i; zero for x < i in BOTH halves (psmnet_3.py:149-163)."""
    b, c, h, w = feat_l.shape
    vol = feat_l.new_zeros(b, 2 * c, ndisp, h, w)
    for i in range(ndisp):
        vol[:, :c, i, :, i:] = feat_l[:, :, :, i:]
        vol[:, c:, i, :, i:] = feat_r[:, :, :, : max(w - i, 0)]
    return vol


def soft_argmin_head(cost_lowres, maxdisp, out_h, out_w):
    """[B,1,d,h,w] logits -> [B,1,H,W] expected disparity
    (psmnet_3.py:206-215 + psmnet_submodule_3.py:80-89)."""
    up = F.interpolate(
        cost_lowres, size=(maxdisp, out_h, out_w), mode="trilinear", align_corners=False
    )
    prob = torch.softmax(up[:, 0], dim=1)
    ramp = torch.arange(maxdisp, dtype=prob.dtype, device=prob.device).view(1, -1, 1, 1)
    return (prob * ramp).sum(dim=1, keepdim=True)


def psmnet_disp_loss(preds, gt, mask):
    """0.5*SL1(pred1) + 0.7*SL1(pred2) + SL1(pred3) on masked pixels, mean
    (utils/losses.py:7-15).  preds = (pred3, pred2, pred1)."""
    p3, p2, p1 = preds
    sl1 = lambda p: F.smooth_l1_loss(p[mask], gt[mask], reduction="mean")
    return 0.5 * sl1(p1) + 0.7 * sl1(p2) + sl1(p3)


def disparity_mask(gt, maxdisp):
    """train.py:272 -- valid where 0 < gt < MAX_DISP."""
    return (gt < maxdisp) & (gt > 0)


# --------------------------------------------------------------------------
# module tree (names == reference names)
# --------------------------------------------------------------------------
def _cb2(cin, cout, k, s, pad, dil):
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, s, dil if dil > 1 else pad, dil, bias=False),
        nn.BatchNorm2d(cout),
    )


def _cb3(cin, cout, stride):
    return nn.Sequential(
        nn.Conv3d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm3d(cout)
    )


def _cb3_relu(cin, cout, stride):
    return nn.Sequential(_cb3(cin, cout, stride), nn.ReLU(inplace=True))


def _up3(cin, cout):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False),
        nn.BatchNorm3d(cout),
    )


class _ResBlock2d(nn.Module):
    def __init__(self, cin, cout, stride, shortcut, pad, dil):
        super().__init__()
        self.conv1 = nn.Sequential(_cb2(cin, cout, 3, stride, pad, dil), nn.ReLU(inplace=True))
        self.conv2 = _cb2(cout, cout, 3, 1, pad, dil)
        self.downsample = shortcut

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class FeatureExtractionOracle(nn.Module):
    """psmnet_submodule_3.py:92-220 (in_ch=3) / psmnet_submodule.py (in_ch=6)."""

    def __init__(self, in_ch=3):
        super().__init__()
        relu = lambda: nn.ReLU(inplace=True)
        self.firstconv = nn.Sequential(
            _cb2(in_ch, 32, 3, 2, 1, 1), relu(), _cb2(32, 32, 3, 1, 1, 1), relu(),
            _cb2(32, 32, 3, 1, 1, 1), relu(),
        )
        self._planes = 32
        self.layer1 = self._stage(32, 3, 1, 1, 1)
        self.layer2 = self._stage(64, 16, 2, 1, 1)
        self.layer3 = self._stage(128, 3, 1, 1, 1)
        self.layer4 = self._stage(128, 3, 1, 1, 2)
        for idx, win in ((1, 64), (2, 32), (3, 16), (4, 8)):
            setattr(self, f"branch{idx}", nn.Sequential(
                nn.AvgPool2d((win, win), stride=(win, win)), _cb2(128, 32, 1, 1, 0, 1), relu()))
        self.lastconv = nn.Sequential(
            _cb2(320, 128, 3, 1, 1, 1), relu(), nn.Conv2d(128, 32, 1, 1, 0, bias=False))

    def _stage(self, planes, n, stride, pad, dil):
        sc = None
        if stride != 1 or self._planes != planes:
            sc = nn.Sequential(nn.Conv2d(self._planes, planes, 1, stride, bias=False),
                               nn.BatchNorm2d(planes))
        blocks = [_ResBlock2d(self._planes, planes, stride, sc, pad, dil)]
        self._planes = planes
        blocks += [_ResBlock2d(planes, planes, 1, None, pad, dil) for _ in range(n - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x, x_extra=None):
        if x_extra is not None:  # 6-channel variant: psmnet_submodule.py:172-174
            x = torch.cat((x, x_extra), 1)
        raw = self.layer2(self.layer1(self.firstconv(x)))
        skip = self.layer4(self.layer3(raw))
        hw = skip.shape[-2:]
        pooled = [
            F.interpolate(getattr(self, f"branch{i}")(skip), hw, mode="bilinear", align_corners=True)
            for i in (4, 3, 2, 1)
        ]
        return self.lastconv(torch.cat([raw, skip] + pooled, 1))


class HourglassOracle(nn.Module):
    """psmnet_3.py:11-77."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = _cb3_relu(c, 2 * c, 2)
        self.conv2 = _cb3(2 * c, 2 * c, 1)
        self.conv3 = _cb3_relu(2 * c, 2 * c, 2)
        self.conv4 = _cb3_relu(2 * c, 2 * c, 1)
        self.conv5 = _up3(2 * c, 2 * c)
        self.conv6 = _up3(2 * c, c)

    def forward(self, x, presqu, postqu):
        pre = self.conv2(self.conv1(x))
        pre = F.relu(pre if postqu is None else pre + postqu)
        deep = self.conv4(self.conv3(pre))
        post = F.relu(self.conv5(deep) + (pre if presqu is None else presqu))
        return self.conv6(post), pre, post


class PSMNetOracle(nn.Module):
    """Device-agnostic restatement of reference PSMNet (both variants).

    in_ch=3 -> nets/psmnet/psmnet_3.py ; in_ch=6 -> nets/psmnet/psmnet.py."""

    def __init__(self, maxdisp=192, in_ch=3):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = FeatureExtractionOracle(in_ch)
        self.dres0 = nn.Sequential(_cb3(64, 32, 1), nn.ReLU(inplace=True),
                                   _cb3(32, 32, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(_cb3(32, 32, 1), nn.ReLU(inplace=True), _cb3(32, 32, 1))
        self.dres2 = HourglassOracle(32)
        self.dres3 = HourglassOracle(32)
        self.dres4 = HourglassOracle(32)
        for i in (1, 2, 3):
            setattr(self, f"classif{i}", nn.Sequential(
                _cb3(32, 32, 1), nn.ReLU(inplace=True),
                nn.Conv3d(32, 1, 3, 1, 1, bias=False)))
        reference_init_(self)

    # stage helpers are exposed so tests and bench can time / check them separately
    def aggregate(self, volume):
        c0 = self.dres0(volume)
        c0 = self.dres1(c0) + c0
        o1, pre1, post1 = self.dres2(c0, None, None)
        o1 = o1 + c0
        o2, _pre2, post2 = self.dres3(o1, pre1, post1)
        o2 = o2 + c0
        o3, _pre3, _post3 = self.dres4(o2, pre1, post2)
        o3 = o3 + c0
        k1 = self.classif1(o1)
        k2 = self.classif2(o2) + k1
        k3 = self.classif3(o3) + k2
        return k1, k2, k3

    def forward(self, img_l, img_r, img_l_t=None, img_r_t=None):
        fl = self.feature_extraction(img_l, img_l_t)
        fr = self.feature_extraction(img_r, img_r_t)
        vol = build_cost_volume(fl, fr, self.maxdisp // 4)
        k1, k2, k3 = self.aggregate(vol)
        hh, ww = 4 * fl.shape[2], 4 * fl.shape[3]
        p3 = soft_argmin_head(k3, self.maxdisp, hh, ww)
        if not self.training:
            return p3
        p1 = soft_argmin_head(k1, self.maxdisp, hh, ww)
        p2 = soft_argmin_head(k2, self.maxdisp, hh, ww)
        return p3, p2, p1


def reference_init_(model):
    """psmnet_3.py:123-142: conv ~ N(0, sqrt(2/(k*Cout))), BN gamma=1 beta=0.
    (ConvTranspose3d is not an nn.Conv3d instance and keeps torch's default init.)"""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            n = m.out_channels
            for k in m.kernel_size:
                n *= k
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
