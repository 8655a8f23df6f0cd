"""PSMNet (stacked hourglass) with the MI355X-native hot path, 3-channel input.

Drop-in for the reference module nets/psmnet/psmnet_3.py: same class names
(`hourglass`, `PSMNet`), constructor arguments, forward signature / return
convention (`(pred3, pred2, pred1)` in training, `pred3` in eval, each
[B,1,H,W] fp32) and state-dict keys.  What changes is how forward() computes:

  reference psmnet_3.py:149-163  zeros + 2*D/4 slice copies  -> ops.cost_volume (K3)
  reference psmnet_3.py:165-179  cuDNN conv3d/BN stack       -> activezero_amd.agg3d (K4/K5)
  reference psmnet_3.py:184-215  interpolate+softmax+regress -> ops.softargmin (K6, fused)

The module is device-strict: inputs must be on the GPU (the reference
hard-codes .cuda(); this path has no CPU fallback).
"""
import math

from activezero_amd import overlap
from activezero_amd import agg3d, conv3d, ops
from activezero_amd.nets.psmnet.psmnet_submodule_3 import *  # noqa: F401,F403
from activezero_amd.nets.psmnet import psmnet_submodule_3 as _sub


def _relu_unit(cin, cout, stride):
    return nn.Sequential(convbn_3d(cin, cout, kernel_size=3, stride=stride, pad=1),
                         nn.ReLU(inplace=True))


def _up_unit(cin, cout):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2,
                           bias=False),
        nn.BatchNorm3d(cout))


class hourglass(nn.Module):
    def __init__(self, inplanes):
        super().__init__()
        c2 = inplanes * 2
        self.conv1 = _relu_unit(inplanes, c2, 2)
        self.conv2 = convbn_3d(c2, c2, kernel_size=3, stride=1, pad=1)
        self.conv3 = _relu_unit(c2, c2, 2)
        self.conv4 = _relu_unit(c2, c2, 1)
        self.conv5 = _up_unit(c2, c2)
        self.conv6 = _up_unit(c2, inplanes)

    def forward(self, x, presqu, postqu, out_add=None, arith=None):
        """Reference signature (psmnet_3.py:36) plus `out_add`: a tensor added to `out` inside conv6's
        BatchNorm pass -- PSMNet.forward adds cost0 to every hourglass output (psmnet_3.py:166-175),
        which otherwise is one more read+write of the 32-channel V0 tensor per hourglass -- and
        `arith`: the MFMA arithmetic (conv3d.Arith; None = the library default)."""
        a = arith
        down = agg3d.conv_bn(x, self.conv1[0], relu=True, arith=a)
        pre = agg3d.conv_bn(down, self.conv2, relu=True, add=postqu, arith=a)
        deep = agg3d.conv_bn(pre, self.conv3[0], relu=True, arith=a)
        deep = agg3d.conv_bn(deep, self.conv4[0], relu=True, arith=a)
        post = agg3d.deconv_bn(deep, self.conv5, relu=True,
                               add=pre if presqu is None else presqu, arith=a)
        out = agg3d.deconv_bn(post, self.conv6, add=out_add, arith=a)
        return out, pre, post


def _classifier():
    return nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                         nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False))


class PSMNet(nn.Module):
    _feature_module = _sub

    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = self._feature_module.FeatureExtraction()
        self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif1 = _classifier()
        self.classif2 = _classifier()
        self.classif3 = _classifier()
        self._reference_init()
        # arithmetic of the MFMA kernels of THIS module (conv3d.Arith): an attribute, not a
        # process-wide switch -- two models with different arithmetic can run side by side
        self.arith = conv3d.DEFAULT_ARITH
        # weight-gradient kernels of a training pass on a second stream, beside the rest of the backward pass
        self.wgrad_overlap = True

    def set_weight_grad_overlap(self, enabled=True):
        self.wgrad_overlap = bool(enabled)
        return self

    def set_arithmetic(self, conv="f16x3", wgrad=None):
        """'f16x3' (default: two scaled fp16 parts, three MFMAs per product, where a kernel for the shape exists),
        'bf16x6' (exact 3-way bf16 split, six MFMAs per product) or 'fp32' (fp32 MFMA)."""
        self.arith = conv3d.Arith.of(conv, wgrad)
        return self

    def _reference_init(self):
        # reference psmnet_3.py:123-142
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                fan = m.out_channels * math.prod(m.kernel_size)
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.bias.data.zero_()

    # -- hot path ------------------------------------------------------------
    def _aggregate(self, vol, first=None, arith=None):
        a = self.arith if arith is None else arith
        # `first`: dres0[0]'s activation when it was computed straight from the feature maps
        c0 = first if first is not None else agg3d.conv_bn(vol, self.dres0[0], relu=True, arith=a)
        c0 = agg3d.conv_bn(c0, self.dres0[2], relu=True, arith=a)
        t = agg3d.conv_bn(c0, self.dres1[0], relu=True, arith=a)
        c0 = agg3d.conv_bn(t, self.dres1[2], add=c0, arith=a)

        # cost0 has four consumers (first hourglass + three residual sums): one fused gradient sum
        c0a, c0b, c0c, c0d = agg3d.fanout(c0, 4)
        out1, pre1, post1 = self.dres2(c0a, None, None, out_add=c0b, arith=a)   # out_k + cost0 fused
        out2, _pre2, post2 = self.dres3(out1, pre1, post1, out_add=c0c, arith=a)
        out3, _pre3, _post3 = self.dres4(out2, pre1, post2, out_add=c0d, arith=a)

        def head(cls, v, running):
            # classifN: convbn_3d + ReLU, then Conv3d(32 -> 1) (psmnet_3.py:103-117).  In train mode the BatchNorm
            # apply + ReLU is not a pass of its own: the 32 -> 1 kernels take it while they stage their input
            aff = conv3d.DeferredAffine() if (torch.is_grad_enabled() and cls[0][1].training) else None
            mid = agg3d.conv_bn(v, cls[0], relu=True, arith=a, defer=aff)
            return agg3d.conv_logits(mid, cls[2], running, arith=a, affine=aff)

        cost1 = head(self.classif1, out1, None)
        cost2 = head(self.classif2, out2, cost1)
        cost3 = head(self.classif3, out3, cost2)
        return cost1, cost2, cost3

    def _pass_arith(self, like):
        """The per-call arithmetic of ONE forward pass: self.arith plus, in training passes, a fresh
        overlap.Sink (weight-gradient kernels on a side stream; overlap.py explains why autograd and DDP are
        unaffected).  Nothing is stored on the module."""
        sink = overlap.begin(self, like) if self.wgrad_overlap else None
        if self.arith.conv == conv3d.F16X3 or self.arith.bwd16:
            # the f16x3 kernels scale their weights by max |w|: all of them in one multi-tensor pass per optimizer step.
            # (the gated aliases of a training pass share storage and version counter with the parameters)
            ws = overlap.conv_weights(self)
            if torch.is_grad_enabled():
                conv3d.prepack(ws)  # ... and every packed image the last step asked for, in one more launch (conv3d.PackPlan)
            conv3d.prime_weight_amax(ws)
        return self.arith._replace(sink=sink)

    def _from_features(self, feat_l, feat_r, arith=None):
        a = self.arith if arith is None else arith
        # cost volume + dres0[0] factored into 2-D convolutions of the two feature maps: the
        # [B,64,D/4,h,w] volume (psmnet_3.py:149-163) and its gradient are never formed
        first = agg3d.costvol_conv_bn(feat_l, feat_r, self.maxdisp // 4, self.dres0[0], relu=True, arith=a)
        cost1, cost2, cost3 = self._aggregate(None, first, a)
        pred3 = ops.softargmin(cost3)
        if self.training:
            return pred3, ops.softargmin(cost2), ops.softargmin(cost1)
        return pred3

    @staticmethod
    def _nhwc(img):
        return img.contiguous(memory_format=torch.channels_last)

    def forward(self, img_L, img_R):
        # both images in one pass of the extractor, statistics per image set (forward_pair)
        a = self._pass_arith(img_L)
        return self._from_features(*self.feature_extraction.forward_pair(self._nhwc(img_L), self._nhwc(img_R), a), a)
