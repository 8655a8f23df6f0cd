"""GPU: the batch-walking 3x3 convolution kernel (az_conv2d_roll.hip, include/azhip.h K13r) through its C ABI, against
torch's fp64 convolution: plain, with the fused epilogue (scale / shift / residual / ReLU), as input gradient (flipped
packing), and its BatchNorm partials per statistic group -- ragged images, 32 / 64 channels on either side.
Reference layers: nets/psmnet/psmnet_submodule_3.py:92-147 (firstconv[1..2], layer1, layer2)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import _lib  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402
from tests._weights import seeded  # noqa: E402

DEV = "cuda:0"


def _pack(w, cin, cout, s_out, s_in, flip):
    pk = torch.empty(int(_lib.lib().az_conv2d_roll_packed_floats(cin, cout)), device=DEV)
    _call("az_conv2d_roll_pack", _p(pk), _p(w), cin, cout, s_out, s_in, int(flip), _stream())
    return pk


@pytest.mark.parametrize("b,h,w,cin,cout", [(4, 21, 37, 32, 32), (2, 24, 48, 64, 64), (3, 9, 50, 32, 64), (2, 37, 53, 64, 32),
                                            (8, 64, 80, 32, 32)])
def test_forward_epilogue_and_input_gradient(b, h, w, cin, cout):
    x = seeded((b, h, w, cin), 11).to(DEV)
    wt = (seeded((cout, cin, 3, 3), 12) * 0.1).to(DEV)
    res = seeded((b, h, w, cout), 13).to(DEV)
    sc, sh = (seeded((cout,), 14) * 0.5 + 1.0).to(DEV), seeded((cout,), 15).to(DEV)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    pk = _pack(wt, cin, cout, cin * 9, 9, False)
    out = torch.empty(b, h, w, cout, device=DEV)
    _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), None, None, None, 0, b, h, w, cin, cout, _stream())
    tol = 2e-5 * float(ref.abs().max())  # the tolerance of the other bf16x6 convolution tests
    assert float((out.double() - ref).abs().max()) <= tol
    _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), _p(sc), _p(sh), _p(res), 1, b, h, w, cin, cout, _stream())
    want = F.relu(ref * sc.double() + sh.double() + res.double())
    assert float((out.double() - want).abs().max()) <= 2 * tol
    # input gradient of the layer: the same kernel on the flipped, role-swapped packing
    g = seeded((b, h, w, cout), 16).to(DEV)
    gref = F.conv_transpose2d(g.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    pkd = _pack(wt, cout, cin, 9, cin * 9, True)
    gx = torch.empty(b, h, w, cin, device=DEV)
    _call("az_conv2d_roll_fwd", _p(gx), _p(g), _p(pkd), None, None, None, 0, b, h, w, cout, cin, _stream())
    assert float((gx.double() - gref).abs().max()) <= 2e-5 * float(gref.abs().max())


@pytest.mark.parametrize("b,h,w,cin,cout,groups", [(4, 21, 37, 32, 32, 1), (4, 24, 48, 64, 64, 2), (6, 17, 23, 32, 64, 2),
                                                   (8, 64, 80, 64, 32, 2)])
def test_batchnorm_partials_per_group(b, h, w, cin, cout, groups):
    x = seeded((b, h, w, cin), 21).to(DEV)
    wt = (seeded((cout, cin, 3, 3), 22) * 0.1).to(DEV)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    pk = _pack(wt, cin, cout, cin * 9, 9, False)
    rows = int(_lib.lib().az_conv2d_roll_stats_rows(groups, b, h, w, cin, cout))
    assert rows > 0
    part = torch.full((groups, cout, rows, 2), float("nan"), device=DEV)  # every row must be written
    cnt = torch.full((groups, rows), float("nan"), device=DEV)
    out = torch.empty(b, h, w, cout, device=DEV)
    _call("az_conv2d_roll_fwd_stats", _p(out), _p(part), _p(cnt), _p(x), _p(pk), groups, b, h, w, cin, cout, _stream())
    torch.cuda.synchronize()
    assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert not torch.isnan(part).any() and not torch.isnan(cnt).any()
    n = cnt.double().sum(1)
    assert torch.equal(n.cpu(), torch.full((groups,), float(b // groups * h * w), dtype=torch.float64))
    mean = part[..., 0].double().sum(2) / n[:, None]
    tile_mean = part[..., 0].double() / cnt.double().clamp_min(1.0)[:, None, :]
    m2 = part[..., 1].double().sum(2) + (cnt.double()[:, None, :] * (tile_mean - mean[:, :, None]) ** 2).sum(2)
    rg = ref.view(groups, b // groups, h, w, cout)
    mean_ref, var_ref = rg.mean(dim=(1, 2, 3)), rg.var(dim=(1, 2, 3), unbiased=False)
    assert float(((mean - mean_ref).abs() / var_ref.sqrt()).max()) <= 1e-5
    assert float((m2 / n[:, None] / var_ref - 1).abs().max()) <= 1e-4


def test_unsupported_shapes_are_refused():
    lib = _lib.lib()
    assert lib.az_conv2d_roll_packed_floats(128, 128) < 0 and lib.az_conv2d_roll_packed_floats(32, 96) < 0
    assert lib.az_conv2d_roll_stats_rows(3, 4, 16, 16, 32, 32) < 0  # the batch does not divide into the groups
