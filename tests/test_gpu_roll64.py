"""Round 5: the stride-1 64 -> 64 convolutions (hourglass conv2 / conv4, reference nets/psmnet/psmnet_3.py:23-33, and their
input gradients) on the depth-rolling kernel as two workgroups per patch, one per half of the output channels
(az_conv3d_roll.hip, COUT = 64; weights in the AZ_PACK_3D_ROLL2 layout) -- against torch's fp64 convolution on shapes with
several patches, ragged edges, one plane and depths that split into segments; every epilogue; the pre-split operand."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import _lib, conv3d  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402
from tests.test_gpu_s2roll import _presplit, cl, ncdhw, seeded  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(1, 6, 17, 35), (2, 9, 8, 16), (1, 24, 10, 18), (1, 1, 3, 3), (3, 4, 9, 30), (1, 13, 25, 17)]


def bound(x, w):
    return 4.0 * 2.0 ** -22 * F.conv3d(x.abs().double(), w.abs().double(), padding=1).max().item()


def on_roll64():
    return _lib.lib().az_option(b"AZ_CONV_ROLL64") != 0


def test_routing_and_layout():
    lib = _lib.lib()
    if not on_roll64():
        pytest.skip("AZ_CONV_ROLL64=0")
    assert lib.az_conv3d_f16_layout(conv3d.CONV_S1, 64, 64) == conv3d.PACK_3D_ROLL2
    assert lib.az_conv3d_f16_layout(conv3d.CONV_S1, 64, 32) == conv3d.PACK_3D_ROLL
    assert lib.az_conv3d_f16_layout(conv3d.CONV_S1, 32, 64) == conv3d.PACK_3D_GATHER
    assert lib.az_conv3d_fwd_f16_split_ok(conv3d.CONV_S1, 4, 64, 64, 24, 68, 120) == 1


@pytest.mark.parametrize("dims", SHAPES)
def test_forward_vs_torch_fp64(dims):
    b, d, h, w = dims
    x = seeded((b, 64, d, h, w), 1)
    wt = seeded((64, 64, 3, 3, 3), 2, -0.2, 0.2)
    ref = F.conv3d(x.double(), wt.double(), padding=1)
    out = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S1, conv3d.F16X3)
    assert (ncdhw(out).double() - ref).abs().max().item() <= bound(x, wt)


@pytest.mark.parametrize("dims", SHAPES[:4])
def test_epilogue_scale_shift_residual_relu(dims):
    b, d, h, w = dims
    x = seeded((b, 64, d, h, w), 3)
    wt = seeded((64, 64, 3, 3, 3), 4, -0.2, 0.2)
    sc, sh = seeded((64,), 5, 0.5, 1.5), seeded((64,), 6, -0.5, 0.5)
    conv = F.conv3d(x.double(), wt.double(), padding=1)
    res = seeded(tuple(conv.shape), 7)
    ref = F.relu(conv * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1) + res.double())
    out = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S1, conv3d.F16X3, scale=sc.to(DEV), shift=sh.to(DEV), residual=cl(res), relu=True)
    assert (ncdhw(out).double() - ref).abs().max().item() <= 1.5 * bound(x, wt) + 1e-6
    assert float(out.min()) >= 0.0


@pytest.mark.parametrize("dims", SHAPES)
def test_batchnorm_partials_merge_to_the_moments_of_the_output(dims):
    b, d, h, w = dims
    x = seeded((b, 64, d, h, w), 8)
    wt = seeded((64, 64, 3, 3, 3), 9, -0.2, 0.2)
    raw, part, cnt, ntiles = conv3d._conv(cl(x), wt.to(DEV), conv3d.CONV_S1, conv3d.F16X3, stats=True)
    ref = F.conv3d(x.double(), wt.double(), padding=1)
    assert (ncdhw(raw).double() - ref).abs().max().item() <= bound(x, wt)
    part, cnt = part.double().cpu().numpy(), cnt.double().cpu().numpy()
    assert part.shape == (64, ntiles, 2)
    vox = raw.numel() // 64
    assert cnt.sum() == vox
    y = raw.double().reshape(-1, 64).cpu().numpy()
    mean = part[:, :, 0].sum(1) / vox
    np.testing.assert_allclose(mean, y.mean(0), rtol=1e-5, atol=1e-6)
    live = cnt > 0
    tile_mean = np.where(live, part[:, :, 0] / np.maximum(cnt, 1), 0.0)
    m2 = (part[:, :, 1] + cnt * (tile_mean - mean[:, None]) ** 2 * live).sum(1)
    np.testing.assert_allclose(m2 / vox, y.var(0), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("dims", SHAPES[:5])
@pytest.mark.parametrize("with_residual", [False, True])
def test_input_gradient(dims, with_residual):
    b, d, h, w = dims
    wt = seeded((64, 64, 3, 3, 3), 10, -0.2, 0.2)
    xin = seeded((b, 64, d, h, w), 11).double().requires_grad_()
    dy = seeded((b, 64, d, h, w), 12) * 1e-3
    F.conv3d(xin, wt.double(), padding=1).backward(dy.double())
    res = seeded(tuple(xin.shape), 13) * 1e-3 if with_residual else None
    got = conv3d._input_grad(cl(dy), wt.to(DEV), conv3d.CONV_S1, 64, 64, conv3d.F16X3, residual=None if res is None else cl(res))
    ref = xin.grad + (res.double() if with_residual else 0.0)
    tol = 4.0 * 2.0 ** -22 * F.conv_transpose3d(dy.abs().double(), wt.abs().double(), padding=1).max().item()
    assert (ncdhw(got).double() - ref).abs().max().item() <= tol + 1e-12


@pytest.mark.parametrize("dims", [(1, 6, 17, 35), (2, 8, 8, 16), (1, 24, 10, 18)])
def test_presplit_gradient_operand(dims):
    if not conv3d.PRESPLIT or not on_roll64():
        pytest.skip("AZ_PRESPLIT=0 / AZ_CONV_ROLL64=0")
    b, d, h, w = dims
    wt = seeded((64, 64, 3, 3, 3), 15, -0.2, 0.2).to(DEV)
    dy = cl(seeded((b, 64, d, h, w), 16) * 1e-3)
    split, plain = _presplit(dy)
    got = conv3d._input_grad(split, wt, conv3d.CONV_S1, 64, 64, conv3d.F16X3)
    want = conv3d._input_grad(plain, wt, conv3d.CONV_S1, 64, 64, conv3d.F16X3)
    ref = F.conv_transpose3d(ncdhw(plain).double(), wt.cpu().double(), padding=1)
    tol = 4.0 * 2.0 ** -22 * F.conv_transpose3d(ncdhw(plain).abs().double(), wt.cpu().abs().double(), padding=1).max().item()
    assert (ncdhw(got).double() - ref).abs().max().item() <= tol
    assert (got - want).abs().max().item() <= tol
