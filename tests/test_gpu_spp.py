"""GPU: the SPP branch upsampling written straight into the concat buffer (az_spp.hip; reference
nets/psmnet/psmnet_submodule_3.py:198-211: F.upsample(..., mode="bilinear") of the four pooled maps, then torch.cat) against
torch's own bilinear interpolation with align_corners=True in fp64, forward and adjoint, and the concat node against
torch.cat of the same pieces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd.nets.psmnet import psmnet_submodule_3 as sub  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402
from tests._weights import seeded  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("src,dst", [((2, 3), (136, 240)), ((4, 7), (136, 240)), ((8, 15), (136, 240)), ((17, 30), (136, 240)),
                                     ((1, 1), (8, 8)), ((3, 5), (3, 5)), ((5, 4), (17, 9))])
@pytest.mark.parametrize("c", [32, 16])
def test_upsample_forward_and_adjoint_vs_torch_fp64(src, dst, c):
    b, (hs, ws), (h, w) = 2, src, dst
    x = seeded((b, c, hs, ws), 70)
    ref_in = x.double().requires_grad_()
    ref = F.interpolate(ref_in, size=(h, w), mode="bilinear", align_corners=True)
    cot = seeded((b, c, h, w), 71)
    ref.backward(cot.double())
    ctot, at = 96, 40  # the branch's slot inside a wider row, as in the concat buffer
    out = torch.full((b, h, w, ctot), 7.0, device=DEV)
    rows = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    _call("az_spp_upsample_fwd", _p(out[..., at:]), _p(rows), b, hs, ws, h, w, c, ctot, _stream())
    got = out[..., at:at + c].permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(got.double(), ref.detach(), rtol=0, atol=1e-5 * float(ref.detach().abs().max()))  # (fp32 source positions, as ATen computes them, against fp64 ones)
    assert float((out[..., :at] - 7.0).abs().max()) == 0.0 and float((out[..., at + c:] - 7.0).abs().max()) == 0.0
    gout = torch.zeros(b, h, w, ctot, device=DEV)
    gout[..., at:at + c] = cot.permute(0, 2, 3, 1).to(DEV)
    gin = torch.empty(b, hs, ws, c, device=DEV)
    from activezero_amd import _lib
    wsb = _lib.lib().az_spp_upsample_bwd_workspace(b, ws, h, c)
    wsp = torch.empty(wsb // 4, device=DEV)
    _call("az_spp_upsample_bwd", _p(gin), _p(wsp), wsb, _p(gout[..., at:]), b, hs, ws, h, w, c, ctot, _stream())
    torch.testing.assert_close(gin.permute(0, 3, 1, 2).cpu().double(), ref_in.grad, rtol=0, atol=2e-5 * float(ref_in.grad.abs().max()))


def test_concat_node_equals_cat_of_interpolations():
    b, h, w = 2, 24, 40
    raw = seeded((b, 64, h, w), 72).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    skip = seeded((b, 128, h, w), 73).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    br = [seeded((b, 32, hs, ws), 74 + i).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
          for i, (hs, ws) in enumerate([(3, 5), (1, 2), (6, 10), (12, 20)])]
    y = sub.spp_concat(raw, skip, br)
    assert y.shape == (b, 320, h, w) and y.is_contiguous(memory_format=torch.channels_last)
    cot = seeded((b, 320, h, w), 80).to(DEV)
    y.backward(cot)
    got = [t.grad.clone() for t in [raw, skip] + br]
    for t in [raw, skip] + br:
        t.grad = None
    ref = torch.cat([raw, skip] + [F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True) for t in br], 1)
    torch.testing.assert_close(y, ref, rtol=0, atol=1e-5 * float(ref.detach().abs().max()))  # (fp32 source positions, as ATen computes them, against fp64 ones)
    ref.backward(cot)
    for a, t in zip(got, [raw, skip] + br):
        torch.testing.assert_close(a, t.grad, rtol=0, atol=2e-5 * float(t.grad.abs().max()))
