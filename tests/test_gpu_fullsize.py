"""GPU, BASELINE.json full sizes (544x960, D=192 -> V0 = 48x136x240): size-independent
properties of the 3-D kernels where the CPU oracle would take minutes.
  * adjointness:  <conv(x), y> == <x, dgrad(y)>   (forward kernel vs input-gradient kernel)
  * bilinearity:  <conv_w(x), y> == <w, wgrad(x, y)>
  * train-mode BatchNorm: the normalised output has per-channel mean 0 / variance 1
  * the fused cost-volume operand equals the materialised one (to rounding: two kernels)
One sample (B=1) keeps the memory footprint at a few GB."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from activezero_amd import agg3d, conv3d, ops  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_3  # noqa: E402

DEV = "cuda:0"
V0 = (1, 48, 136, 240)


def _rand(*shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, device=DEV, generator=g)


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def _same(lhs, rhs, a, b, rel=1e-6):
    """<a,b>-type identities: the inner products cancel heavily (result ~1e3 from 5e7 terms
    of size ~1), so the error bound is relative to |a|*|b| (Cauchy-Schwarz scale), not to the
    value of the sum."""
    scale = a.double().norm().item() * b.double().norm().item()
    assert abs(lhs - rhs) <= rel * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("cin,cout,stride,dims", [
    (32, 32, 1, V0), (64, 32, 1, V0), (32, 64, 2, V0), (64, 64, 2, (1, 24, 68, 120)),
    (64, 64, 1, (1, 24, 68, 120))])
def test_conv_adjoint_and_bilinear_identities_full_size(cin, cout, stride, dims):
    unit = psmnet_3.convbn_3d(cin, cout, 3, stride, 1).to(DEV).train()
    x = _rand(*dims, cin, seed=1).requires_grad_()
    y = agg3d.conv_bn(x, unit)          # conv + train-mode BN
    raw_shape = y.shape
    # BN property: with gamma=1, beta=0 (reference init) y itself is the normalised tensor
    m = y.detach().double().mean(dim=(0, 1, 2, 3))
    v = y.detach().double().var(dim=(0, 1, 2, 3), unbiased=False)
    assert m.abs().max().item() < 1e-4
    assert (v - 1).abs().max().item() < 1e-3
    # identities on the bare convolution kernels
    w = unit[0].weight.detach()
    mode = conv3d.CONV_S1 if stride == 1 else conv3d.CONV_S2
    xd = x.detach()
    out = conv3d.conv_plain(xd, w, mode)
    assert out.shape == raw_shape
    cot = _rand(*out.shape, seed=2)
    bwd = conv3d.F16X3 if conv3d.DEFAULT_ARITH.bwd16 else conv3d.DEFAULT_ARITH.conv
    gx = conv3d._input_grad(cot, w, mode, cin, cout, bwd)
    lhs, rhs = _dot(out, cot), _dot(xd, gx)
    _same(lhs, rhs, out, cot)
    gw = conv3d._weight_grad(xd, cot, mode, cin, cout, conv3d.F16X3 if conv3d.DEFAULT_ARITH.bwd16 else conv3d.DEFAULT_ARITH.wgrad)
    _same(lhs, _dot(w, gw), out, cot)


def test_deconv_identities_full_size():
    cin, cout, dims = 64, 32, (1, 24, 68, 120)
    unit = psmnet_3._up_unit(cin, cout).to(DEV)
    w = unit[0].weight.detach()
    x = _rand(*dims, cin, seed=3)
    out = conv3d.conv_plain(x, w, conv3d.DECONV_S2)
    assert out.shape == (1, 48, 136, 240, cout)
    cot = _rand(*out.shape, seed=4)
    bwd = conv3d.F16X3 if conv3d.DEFAULT_ARITH.bwd16 else conv3d.DEFAULT_ARITH.conv
    gx = conv3d._input_grad(cot, w, conv3d.DECONV_S2, cin, cout, bwd)
    lhs, rhs = _dot(out, cot), _dot(x, gx)
    _same(lhs, rhs, out, cot)
    gw = conv3d._weight_grad(x, cot, conv3d.DECONV_S2, cin, cout, conv3d.F16X3 if conv3d.DEFAULT_ARITH.bwd16 else conv3d.DEFAULT_ARITH.wgrad)
    _same(lhs, _dot(w, gw), out, cot)


def test_fused_cost_volume_full_size_equals_materialised():
    fl, fr = _rand(1, 136, 240, 32, seed=5), _rand(1, 136, 240, 32, seed=6)
    unit = psmnet_3.convbn_3d(64, 32, 3, 1, 1).to(DEV).eval()
    with torch.no_grad():
        y1 = conv3d.conv_bn(conv3d.LazyCostVolume(fl, fr, 48), unit[0], unit[1], conv3d.CONV_S1, True)
        y2 = conv3d.conv_bn(ops.cost_volume_ndhwc(fl, fr, 48), unit[0], unit[1], conv3d.CONV_S1, True)
    # (bit-equal while both ran on one kernel; since round 3 the materialised volume goes through the depth-rolling
    #  16x16x32 kernel and the fused operand through the 32x32x16 one: same arithmetic, K blocks of 32 vs 16)
    assert torch.allclose(y1, y2, rtol=1e-5, atol=2e-6 * float(y2.abs().max()))


def test_classifier_identities_full_size():
    conv = torch.nn.Conv3d(32, 1, 3, padding=1, bias=False).to(DEV)
    x = _rand(*V0, 32, seed=7).requires_grad_()
    y = conv3d.conv_logits(x, conv, None)
    cot = _rand(*y.shape, seed=8)
    y.backward(cot)
    lhs = _dot(y.detach(), cot)
    _same(lhs, _dot(x.detach(), x.grad), y.detach(), cot)
    # <w, gw> is a sum of 864 terms ~500x larger than the result: bound the error by the
    # magnitude of the terms, not of the (heavily cancelling) sum
    wg = conv.weight.detach().double() * conv.weight.grad.double()
    assert abs(lhs - wg.sum().item()) <= 1e-6 * wg.abs().sum().item()
