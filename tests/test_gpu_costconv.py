"""GPU parity of the factored cost-volume convolution (activezero_amd/costconv.py + az_costconv_assemble_*):
conv3d(concat cost volume) computed from the two feature maps without the volume, against the oracle's
materialised volume + F.conv3d -- output and all three gradients, including the staircase mask
(x >= d), both depth borders, the image's right edge, D = 1, 2 and D > W."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import agg3d, costconv  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import seeded  # noqa: E402

DEV = "cuda:0"


def _rel(a, r):
    a, r = a.detach().cpu(), r.detach()
    return float((a - r).abs().max() / (r.abs().max() + 1e-12))


@pytest.mark.parametrize("dims", [(1, 5, 9, 4), (2, 6, 12, 3), (1, 4, 7, 2), (1, 3, 6, 1), (1, 4, 5, 8), (2, 8, 40, 12)])
def test_costvol_conv_matches_materialised_volume(dims):
    b, h, w, nd = dims
    fl = seeded((b, 32, h, w), 11).requires_grad_()
    fr = seeded((b, 32, h, w), 12).requires_grad_()
    wt = seeded((32, 64, 3, 3, 3), 13, -0.1, 0.1).requires_grad_()
    ref = F.conv3d(po.build_cost_volume(fl, fr, nd), wt, padding=1)  # [B,32,nd,h,w]
    ct = seeded(tuple(ref.shape), 14)
    gl, gr, gw = torch.autograd.grad(ref, (fl, fr, wt), ct)
    fl2, fr2, wt2 = (t.detach().to(DEV).requires_grad_() for t in (fl, fr, wt))
    out = costconv.costvol_conv(fl2, fr2, nd, wt2).permute(0, 4, 1, 2, 3)  # NDHWC -> NCDHW view
    assert _rel(out, ref) < 5e-6
    g2 = torch.autograd.grad(out, (fl2, fr2, wt2), ct.to(DEV))
    assert _rel(g2[0], gl) < 1e-5 and _rel(g2[1], gr) < 1e-5 and _rel(g2[2], gw) < 1e-5


def test_costvol_conv_bn_equals_volume_path():
    """The factored conv + BatchNorm3d + ReLU unit against the materialised-volume unit (K3 + the
    64 -> 32 3-D convolution): activation, feature gradients, parameter gradients and running statistics."""
    import copy

    from activezero_amd.nets.psmnet import psmnet_submodule_3 as sm

    torch.manual_seed(0)
    unit_a = sm.convbn_3d(64, 32, 3, 1, 1).to(DEV).train()
    unit_b = copy.deepcopy(unit_a)
    fl = seeded((2, 32, 8, 20), 21).to(DEV).requires_grad_()
    fr = seeded((2, 32, 8, 20), 22).to(DEV).requires_grad_()
    ct = seeded((2, 6, 8, 20, 32), 23).to(DEV)
    ya = agg3d.costvol_conv_bn(fl, fr, 6, unit_a, relu=True)
    ga = torch.autograd.grad(ya, (fl, fr, unit_a[0].weight, unit_a[1].weight, unit_a[1].bias), ct)
    fl2, fr2 = fl.detach().clone().requires_grad_(), fr.detach().clone().requires_grad_()
    vol = agg3d.volume_from_features(fl2, fr2, 6, lazy=False)
    yb = agg3d.conv_bn(vol, unit_b, relu=True)
    gb = torch.autograd.grad(yb, (fl2, fr2, unit_b[0].weight, unit_b[1].weight, unit_b[1].bias), ct)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-5)
    for a, b_ in zip(ga, gb):
        assert torch.allclose(a, b_, rtol=1e-3, atol=2e-4 * float(b_.abs().max()))
    assert torch.allclose(unit_a[1].running_var, unit_b[1].running_var, rtol=1e-5, atol=1e-6)
    assert int(unit_a[1].num_batches_tracked) == int(unit_b[1].num_batches_tracked) == 1


def test_merged_kernels_on_own_kernels_equal_the_einsums():
    """az_costconv_merge_fwd / _bwd (round 5: the step's last rocBLAS launches were these two weight-space einsums) against
    torch.einsum in fp64, values and the gradient of the Conv3d weight, for every depth-class count"""
    from activezero_amd import costconv
    for nd in (1, 2, 6):
        w = seeded((32, 64, 3, 3, 3), 51 + nd, -0.3, 0.3).to(DEV).requires_grad_()
        ml, mr = costconv._masks(w.device, nd)
        ncls = costconv.num_classes(nd)
        kl, kr = costconv._Merge.apply(w, ml, mr, ncls)
        wd = w.detach().double().requires_grad_()
        rl = torch.einsum("oidhw,cedw->ceoihw", wd[:, :32], ml.double())
        rr = torch.einsum("oidhw,cedwj->ceoihj", wd[:, 32:], mr.double())
        torch.testing.assert_close(kl.double(), rl, rtol=0, atol=1e-6)
        torch.testing.assert_close(kr.double(), rr, rtol=0, atol=1e-6)
        cl_, cr_ = seeded(tuple(kl.shape), 60).to(DEV), seeded(tuple(kr.shape), 61).to(DEV)
        ((kl * cl_).sum() + (kr * cr_).sum()).backward()
        ((rl * cl_.double()).sum() + (rr * cr_.double()).sum()).backward()
        torch.testing.assert_close(w.grad.double(), wd.grad, rtol=0, atol=1e-5)
