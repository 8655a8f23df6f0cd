"""Round 5: the stride-2 32 -> 64 convolution on depth-rolling workgroups (az_conv3d_s2roll.hip): hourglass conv1 forward
(reference nets/psmnet/psmnet_3.py:15-22) and the input gradient of the transposed conv6 (psmnet_3.py:34-58), against torch's
fp64 convolution -- shapes with several patches in both directions, ragged patches, odd fine sizes, one plane, and depths
that split into several segments; every epilogue (scale / shift / residual / ReLU, BatchNorm partials), and the pre-split
operand a BatchNorm backward hands to the input gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import _lib, conv3d  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402

DEV = torch.device("cuda:0")

SHAPES = [(1, 6, 34, 70), (2, 9, 16, 32), (1, 48, 20, 36), (1, 1, 3, 3), (1, 7, 33, 65), (3, 4, 18, 30), (1, 13, 50, 34)]


def seeded(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def ncdhw(y):
    return y.permute(0, 4, 1, 2, 3).cpu()


def bound(x, w, stride_conv=True):
    """the f16x3 contract (include/azhip.h): a few 2^-22 of sum |x||w| per output; here its largest value"""
    ax, aw = x.abs().double(), w.abs().double()
    s = F.conv3d(ax, aw, stride=2, padding=1) if stride_conv else F.conv_transpose3d(ax, aw, stride=2, padding=1, output_padding=1)
    return 4.0 * 2.0 ** -22 * s.max().item()


def on_s2roll():
    return _lib.lib().az_option(b"AZ_CONV_S2ROLL") != 0


def test_the_layers_are_routed_to_the_rolling_kernel():
    lib = _lib.lib()
    if not on_s2roll():
        pytest.skip("AZ_CONV_S2ROLL=0")
    assert lib.az_conv3d_f16_layout(conv3d.CONV_S2, 32, 64) == lib.az_conv3d_f16_layout(conv3d.CONV_S1, 32, 32)  # the rolling layout
    assert lib.az_conv3d_f16_layout(conv3d.CONV_S2, 64, 64) != lib.az_conv3d_f16_layout(conv3d.CONV_S1, 32, 32)
    assert lib.az_conv3d_fwd_f16_split_ok(conv3d.CONV_S2, 4, 32, 64, 48, 136, 240) == 1
    # one row of BatchNorm partials per workgroup: far fewer than the gather kernel's one per 4 x 8 tile and plane
    assert lib.az_conv3d_stats_tiles_f16(conv3d.CONV_S2, 4, 32, 64, 48, 136, 240) < 4 * 24 * 17 * 15


@pytest.mark.parametrize("dims", SHAPES)
def test_forward_vs_torch_fp64(dims):
    b, d, h, w = dims
    x = seeded((b, 32, d, h, w), 1)
    wt = seeded((64, 32, 3, 3, 3), 2, -0.2, 0.2)
    ref = F.conv3d(x.double(), wt.double(), stride=2, padding=1)
    out = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S2, conv3d.F16X3)
    err = (ncdhw(out).double() - ref).abs().max().item()
    assert err <= bound(x, wt), (err, bound(x, wt))


@pytest.mark.parametrize("dims", SHAPES[:4])
def test_epilogue_scale_shift_residual_relu(dims):
    b, d, h, w = dims
    x = seeded((b, 32, d, h, w), 3)
    wt = seeded((64, 32, 3, 3, 3), 4, -0.2, 0.2)
    sc, sh = seeded((64,), 5, 0.5, 1.5), seeded((64,), 6, -0.5, 0.5)
    conv = F.conv3d(x.double(), wt.double(), stride=2, padding=1)
    res = seeded(tuple(conv.shape), 7)
    ref = F.relu(conv * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1) + res.double())
    out = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S2, conv3d.F16X3, scale=sc.to(DEV), shift=sh.to(DEV), residual=cl(res), relu=True)
    err = (ncdhw(out).double() - ref).abs().max().item()
    assert err <= 1.5 * bound(x, wt) + 1e-6, err
    assert float(out.min()) >= 0.0


@pytest.mark.parametrize("dims", SHAPES)
def test_batchnorm_partials_merge_to_the_moments_of_the_output(dims):
    b, d, h, w = dims
    x = seeded((b, 32, d, h, w), 8)
    wt = seeded((64, 32, 3, 3, 3), 9, -0.2, 0.2)
    raw, part, cnt, ntiles = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S2, conv3d.F16X3, stats=True)
    ref = F.conv3d(x.double(), wt.double(), stride=2, padding=1)
    assert (ncdhw(raw).double() - ref).abs().max().item() <= bound(x, wt)
    part, cnt = part.double().cpu().numpy(), cnt.double().cpu().numpy()
    assert part.shape == (64, ntiles, 2)
    vox = raw.numel() // 64
    assert cnt.sum() == vox  # every output voxel counted once
    y = raw.double().reshape(-1, 64).cpu().numpy()
    mean = part[:, :, 0].sum(1) / vox
    np.testing.assert_allclose(mean, y.mean(0), rtol=1e-5, atol=1e-6)
    live = cnt > 0
    tile_mean = np.where(live, part[:, :, 0] / np.maximum(cnt, 1), 0.0)
    m2 = (part[:, :, 1] + cnt * (tile_mean - mean[:, None]) ** 2 * live).sum(1)  # Chan's merge, as az_bn3d_finalize
    np.testing.assert_allclose(m2 / vox, y.var(0), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("dims", SHAPES[:5])
@pytest.mark.parametrize("with_residual", [False, True])
def test_input_gradient_of_the_transposed_layer(dims, with_residual):
    """conv6: y = ConvTranspose3d(x), x: 64 channels coarse, y: 32 channels fine; dx = stride-2 convolution of dy"""
    b, d, h, w = dims
    if d % 2 or h % 2 or w % 2:
        pytest.skip("output_padding = 1 always gives even fine sizes")
    wt = seeded((64, 32, 3, 3, 3), 10, -0.2, 0.2)
    xc = seeded((b, 64, d // 2, h // 2, w // 2), 11).double().requires_grad_()
    dy = seeded((b, 32, d, h, w), 12) * 1e-3
    F.conv_transpose3d(xc, wt.double(), stride=2, padding=1, output_padding=1).backward(dy.double())
    res = seeded(tuple(xc.shape), 13) * 1e-3 if with_residual else None
    got = conv3d._input_grad(cl(dy), wt.to(DEV), conv3d.DECONV_S2, 64, 32, conv3d.F16X3, residual=None if res is None else cl(res))
    ref = xc.grad + (res.double() if with_residual else 0.0)
    w_as_conv = wt  # [ci = 64][co = 32]: the stride-2 convolution of dy reads it as [out 64][in 32]
    err = (ncdhw(got).double() - ref).abs().max().item()
    assert err <= bound(dy, w_as_conv) + 1e-12, (err, bound(dy, w_as_conv))


def _presplit(dy):
    """dy -> the pre-split tensor az_bn3d_bwd(split_out = 1) writes for an identity BatchNorm (mean 0, invstd 1, gamma 1):
    dx = dy - mean(dy) - xhat mean(dy xhat), and that dx as plain floats from a second launch"""
    c = dy.shape[-1]
    nv = dy.numel() // c
    raw = cl(seeded((dy.shape[0], c) + tuple(dy.shape[1:4]), 14))
    wsb = _lib.lib().az_bn3d_bwd_workspace(nv, c)
    v = [torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.ones(c, device=DEV)]
    outs = []
    for split in (1, 0):
        ws, dx = torch.empty(wsb // 4, device=DEV), torch.empty_like(dy)
        small = [torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(c, 3, device=DEV)]
        am = torch.zeros(conv3d.AMAX_SLOTS, device=DEV)
        _call("az_bn3d_bwd", _p(dx), None, _p(small[0]), _p(small[1]), _p(small[2]), _p(ws), wsb, _p(dy), None, _p(raw), _p(v[0]),
              _p(v[1]), _p(v[2]), None, None, 0, nv, c, _p(am), split, _stream())
        if split:
            conv3d._set_amax(dx, am)
            dx.az_split = True
        outs.append(dx)
    return outs


@pytest.mark.parametrize("dims", [(1, 6, 34, 70), (2, 8, 16, 32), (1, 48, 20, 36)])
def test_presplit_gradient_operand(dims):
    if not conv3d.PRESPLIT or not on_s2roll():
        pytest.skip("AZ_PRESPLIT=0 / AZ_CONV_S2ROLL=0")
    b, d, h, w = dims
    wt = seeded((64, 32, 3, 3, 3), 15, -0.2, 0.2).to(DEV)
    dy = cl(seeded((b, 32, d, h, w), 16) * 1e-3)
    split, plain = _presplit(dy)
    got = conv3d._input_grad(split, wt, conv3d.DECONV_S2, 64, 32, conv3d.F16X3)
    want = conv3d._input_grad(plain, wt, conv3d.DECONV_S2, 64, 32, conv3d.F16X3)
    ref = F.conv3d(ncdhw(plain).double(), wt.cpu().double(), stride=2, padding=1)
    tol = bound(ncdhw(plain), wt.cpu())
    assert (ncdhw(got).double() - ref).abs().max().item() <= tol
    # the staged bits differ only through the operand scale (the bound of the pre-split tensor vs the amax of the plain one)
    assert (got - want).abs().max().item() <= tol


def test_full_size_sampled_outputs():
    """B = 4 hourglass size: finite everywhere, and a sampled set of outputs (corners, patch seams, segment seams) against fp64"""
    torch.manual_seed(0)
    b, d, h, w = 4, 48, 136, 240
    wt = (torch.rand(64, 32, 3, 3, 3) * 0.4 - 0.2)
    x1 = torch.rand(b, d, h, w, 32, device=DEV) * 2 - 1
    y1 = conv3d._conv(x1, wt.to(DEV), conv3d.CONV_S2, conv3d.F16X3)
    assert y1.shape == (b, 24, 68, 120, 64)
    assert torch.isfinite(y1).all()
    g = torch.Generator().manual_seed(1)
    pts = [(int(torch.randint(0, b, (1,), generator=g)), int(torch.randint(0, 24, (1,), generator=g)),
            int(torch.randint(0, 68, (1,), generator=g)), int(torch.randint(0, 120, (1,), generator=g))) for _ in range(48)]
    pts += [(0, 0, 0, 0), (3, 23, 67, 119), (1, 0, 67, 0), (2, 23, 0, 119), (0, 11, 63, 111), (3, 12, 64, 112)]
    xp = F.pad(x1, (0, 0, 1, 1, 1, 1, 1, 1)).cpu().double()
    wd = wt.double()
    worst = 0.0
    for (bi, t, y, x) in pts:
        patch = xp[bi, 2 * t:2 * t + 3, 2 * y:2 * y + 3, 2 * x:2 * x + 3, :]  # [kd][kh][kw][ci]
        ref = torch.einsum("dhwc,ocdhw->o", patch, wd)
        worst = max(worst, (y1[bi, t, y, x].cpu().double() - ref).abs().max().item())
    assert worst <= 4.0 * 2.0 ** -22 * 27 * 32 * 0.2, worst
