"""GPU parity: the 2-D convolution family (K13: az_conv2d_fwd / az_conv2d_wgrad / az_im2col_s2k3, and the
stride-2 routes through the 3-D kernels) against torch's CPU conv2d in fp64, for EVERY geometry the
feature extractor (reference nets/psmnet/psmnet_submodule_3.py:92-220) and the factored cost-volume
convolution route to it: forward, input gradient, weight gradient; the eval-mode fused
conv+BN(+residual)(+ReLU) epilogue; and whole conv+BN units against plain torch modules."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import bn2d, conv2d  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_submodule_3 as sm  # noqa: E402
from tests._weights import load_procedural, seeded  # noqa: E402

DEV = "cuda:0"
CL = torch.channels_last


def close(a, b, rtol, atol):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def ref_conv(x, w, stride, pad, dil):
    """fp64 reference: output, input gradient, weight gradient for cotangent ct"""
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    y = F.conv2d(xd, wd, None, stride, pad, dil)
    return xd, wd, y


# every stride-1 "same" geometry routed to az_conv2d_*: (cin, cout, (kh, kw), dilation)
SAME = [(32, 32, (3, 3), 1), (64, 64, (3, 3), 1), (64, 128, (3, 3), 1), (128, 128, (3, 3), 1),
        (128, 128, (3, 3), 2), (320, 128, (3, 3), 1), (64, 128, (1, 1), 1), (128, 32, (1, 1), 1),
        (32, 96, (3, 3), 1), (32, 384, (3, 3), 1), (32, 192, (3, 5), 1)]


@pytest.mark.parametrize("cin,cout,k,dil", SAME)
@pytest.mark.parametrize("dims", [(2, 13, 22), (1, 16, 32), (3, 5, 47)])
def test_conv_same_vs_torch_fp64(cin, cout, k, dil, dims):
    b, h, w = dims
    x = seeded((b, cin, h, w), 11)
    wt = seeded((cout, cin) + k, 12, -0.2, 0.2)
    ct = seeded((b, cout, h, w), 13)
    pad = (dil * (k[0] - 1) // 2, dil * (k[1] - 1) // 2)
    xd, wd, yd = ref_conv(x, wt, 1, pad, dil)
    yd.backward(ct.double())
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    wg = wt.to(DEV).requires_grad_()
    y = conv2d.conv_same(xg, wg, dil)
    assert y.shape == yd.shape
    # bf16x6 products carry ~1e-7 relative error each: scale the absolute bound with the output's magnitude
    s = float(yd.detach().abs().max())
    close(y, yd, 1e-5, 2e-6 * s)
    y.backward(ct.to(DEV))
    close(xg.grad, xd.grad, 1e-5, 2e-6 * float(xd.grad.abs().max()))
    close(wg.grad, wd.grad, 1e-5, 2e-6 * float(wd.grad.abs().max()))


@pytest.mark.parametrize("cin,cout,k,stride", [(32, 64, 3, 2), (32, 64, 1, 2), (3, 32, 3, 2), (6, 32, 3, 2)])
@pytest.mark.parametrize("dims", [(2, 16, 24), (1, 32, 48)])
def test_conv_stride2_routes_vs_torch_fp64(cin, cout, k, stride, dims):
    """layer2.0.conv1 (3-D gather kernels, depth-1 volume), layer2.0.downsample (subsample + 1x1) and
    firstconv.0 (patch extraction + 1x1), forward and all gradients."""
    b, h, w = dims
    m = torch.nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False)
    with torch.no_grad():
        m.weight.copy_(seeded(tuple(m.weight.shape), 21, -0.3, 0.3))
    x = seeded((b, cin, h, w), 22)
    xd, wd, yd = ref_conv(x, m.weight.detach(), stride, (k - 1) // 2, 1)
    ct = seeded(tuple(yd.shape), 23)
    yd.backward(ct.double())
    m = m.to(DEV)
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    y = conv2d.conv(xg, m)
    assert y.shape == yd.shape
    close(y, yd, 1e-5, 3e-6 * float(yd.abs().max()))
    y.backward(ct.to(DEV))
    close(xg.grad, xd.grad, 1e-5, 3e-6 * float(xd.grad.abs().max()))
    close(m.weight.grad, wd.grad, 1e-5, 3e-6 * float(wd.grad.abs().max()))


@pytest.mark.parametrize("c,dil", [(32, 1), (64, 1), (128, 2)])
def test_conv_same_skip_adds_the_shortcut_gradient_in_the_dgrad_epilogue(c, dil):
    """(conv(x), x) with both outputs used, as a residual block does (psmnet_submodule_3.py:69-78): the
    gradient of x is dgrad(gy) + g_shortcut, summed inside the input-gradient kernel."""
    x, w = seeded((2, c, 21, 37), 11), seeded((c, c, 3, 3), 12) * 0.1
    ct, cs = seeded((2, c, 21, 37), 13), seeded((2, c, 21, 37), 14)
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    yd = F.conv2d(xd, wd, None, 1, dil, dil)
    ((yd * ct.double()).sum() + (xd * cs.double()).sum()).backward()
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    wg = w.to(DEV).requires_grad_()
    y, xs = conv2d.conv_same_skip(xg, wg, dil)
    assert xs.data_ptr() == xg.data_ptr()  # an alias, not a copy
    ((y * ct.to(DEV)).sum() + (xs * cs.to(DEV)).sum()).backward()
    close(y, yd, 1e-5, 3e-6 * float(yd.detach().abs().max()))
    close(xg.grad, xd.grad, 1e-5, 3e-6 * float(xd.grad.abs().max()))
    close(wg.grad, wd.grad, 1e-5, 3e-6 * float(wd.grad.abs().max()))
    # one output unused: its gradient arrives as None and the other path is unchanged
    xg.grad = None
    y, xs = conv2d.conv_same_skip(xg, wg, dil)
    (xs * cs.to(DEV)).sum().backward()
    close(xg.grad, cs, 0, 0)
    xg.grad = None
    y, xs = conv2d.conv_same_skip(xg, wg, dil)
    (y * ct.to(DEV)).sum().backward()
    close(xg.grad, xd.grad - cs.double(), 1e-5, 3e-6 * float(xd.grad.abs().max()))


def test_conv_unsupported_geometry_raises():
    m = torch.nn.Conv2d(32, 32, 5, 1, 2, bias=False).to(DEV)
    with pytest.raises(RuntimeError):
        conv2d.conv(torch.zeros(1, 32, 8, 8, device=DEV), m)
    with pytest.raises(RuntimeError):  # CPU tensors never fall back
        conv2d.conv_same(torch.zeros(1, 32, 8, 8), torch.zeros(32, 32, 3, 3), 1)


@pytest.mark.parametrize("cin,cout,k,dil,stride", [(64, 64, 3, 1, 1), (128, 128, 3, 2, 1), (64, 128, 1, 1, 1),
                                                    (32, 64, 3, 1, 2), (3, 32, 3, 1, 2)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (False, True), (True, True)])
def test_convbn_unit_train_and_eval_vs_torch_modules(cin, cout, k, dil, stride, relu, with_res):
    """_convbn_unit (HIP conv + HIP BatchNorm(+residual)(+ReLU)) against the plain nn.Conv2d +
    nn.BatchNorm2d modules it stands for: train mode forward/backward/running statistics, and the
    eval-mode forward where BatchNorm, residual and ReLU are folded into the conv epilogue."""
    unit = load_procedural(sm.convbn(cin, cout, k, stride, (k - 1) // 2, dil), "t.cb2d.").to(DEV).train()
    ref = load_procedural(sm.convbn(cin, cout, k, stride, (k - 1) // 2, dil), "t.cb2d.").double().train()
    h, w = 20, 36
    x = seeded((2, cin, h, w), 61)
    res = seeded((2, cout, h // stride, w // stride), 62)
    ct = seeded((2, cout, h // stride, w // stride), 63)
    xr, rr = x.double().requires_grad_(), res.double().requires_grad_()
    yr = ref(xr) + (rr if with_res else 0)
    yr = F.relu(yr) if relu else yr
    yr.backward(ct.double())
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    rg = res.to(DEV).contiguous(memory_format=CL).requires_grad_()
    y = sm._convbn_unit(xg, unit, relu=relu, residual=rg if with_res else None)
    close(y, yr, 1e-4, 2e-5)
    y.backward(ct.to(DEV))
    close(xg.grad, xr.grad, 1e-3, 1e-4)
    if with_res:
        close(rg.grad, rr.grad, 1e-5, 1e-6)
    close(unit[0].weight.grad, ref[0].weight.grad, 1e-3, 3e-4)
    close(unit[1].weight.grad, ref[1].weight.grad, 1e-3, 1e-3)
    close(unit[1].bias.grad, ref[1].bias.grad, 1e-3, 1e-3)
    close(unit[1].running_mean, ref[1].running_mean, 1e-5, 1e-6)
    close(unit[1].running_var, ref[1].running_var, 1e-5, 1e-6)
    # eval mode, no_grad: the fused epilogue (stride 1) / conv + apply (stride 2)
    unit.eval(); ref.eval()
    with torch.no_grad():
        ye = sm._convbn_unit(xg.detach(), unit, relu=relu, residual=rg.detach() if with_res else None)
        yre = ref(x.double()) + (res.double() if with_res else 0)
        yre = F.relu(yre) if relu else yre
    close(ye, yre, 1e-4, 2e-5)


@pytest.mark.parametrize("c,k,dil,hw,groups", [(32, 3, 1, (21, 37), 1), (64, 3, 1, (24, 48), 2), (128, 3, 2, (17, 23), 2),
                                              (64, 1, 1, (9, 50), 1)])
def test_batchnorm_partials_from_the_conv_epilogue(c, k, dil, hw, groups):
    """az_conv2d_fwd_stats: the (sum, M2) partials the convolution's epilogue reduces per channel and 8x16 patch
    -- ragged patches at the image border included -- give the BatchNorm the statistics its own pass over the
    tensor gives, per statistic group: same output, same running statistics."""
    h, w = hw
    x = seeded((4, c, h, w), 71).to(DEV).contiguous(memory_format=CL)
    wt = (seeded((c, c, k, k), 72) * 0.1).to(DEV)
    want_stats = bn2d.Partials(groups)
    y = conv2d.conv_same(x, wt, dil, stats=want_stats)
    assert want_stats.ready and want_stats.part.shape[:2] == (groups, c) and want_stats.cnt.shape[0] == groups
    # partials -> per-group mean / variance, against the tensor itself
    yg = y.detach().double().reshape(groups, 4 // groups, c, h * w).permute(0, 2, 1, 3).reshape(groups, c, -1)
    n = want_stats.cnt.double().sum(1)                                    # [groups]
    assert torch.equal(n.cpu(), torch.full((groups,), float(4 // groups * h * w), dtype=torch.float64))
    s = want_stats.part[..., 0].double().sum(2)                           # [groups, c]
    mean = s / n[:, None]
    tile_mean = want_stats.part[..., 0].double() / want_stats.cnt.double().clamp_min(1.0)[:, None, :]  # (rows of wave
    # quarters that lie outside a ragged image carry count 0 and sums 0)
    m2 = (want_stats.part[..., 1].double() + want_stats.cnt.double()[:, None, :] * (tile_mean - mean[..., None]) ** 2).sum(2)
    close(mean, yg.mean(2), 1e-5, 1e-6)
    close(m2 / n[:, None], yg.var(2, unbiased=False), 1e-4, 1e-7)
    # and through the BatchNorm: with the partials vs with its own statistics pass
    bn_a = load_procedural(torch.nn.BatchNorm2d(c), "t.part.").to(DEV).train()
    bn_b = load_procedural(torch.nn.BatchNorm2d(c), "t.part.").to(DEV).train()
    ya = bn2d.bn_act(y, bn_a, relu=True, groups=groups, partials=want_stats)
    yb = bn2d.bn_act(y, bn_b, relu=True, groups=groups)
    close(ya, yb, 1e-5, 1e-6)
    close(bn_a.running_mean, bn_b.running_mean, 1e-6, 1e-7)
    close(bn_a.running_var, bn_b.running_var, 1e-5, 1e-7)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == groups


def test_convbn_unit_eval_mode_backward():
    """frozen-BatchNorm fine-tuning (eval mode under autograd), as the reference's plain modules allow"""
    unit = load_procedural(sm.convbn(64, 64, 3, 1, 1, 1), "t.cb2e.").to(DEV).eval()
    ref = load_procedural(sm.convbn(64, 64, 3, 1, 1, 1), "t.cb2e.").double().eval()
    x, ct = seeded((2, 64, 12, 20), 71), seeded((2, 64, 12, 20), 72)
    xr = x.double().requires_grad_()
    F.relu(ref(xr)).backward(ct.double())
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    y = sm._convbn_unit(xg, unit, relu=True)
    y.backward(ct.to(DEV))
    close(y, F.relu(ref(x.double())), 1e-4, 2e-5)
    close(xg.grad, xr.grad, 1e-3, 1e-4)
    close(unit[0].weight.grad, ref[0].weight.grad, 1e-3, 3e-4)
    close(unit[1].weight.grad, ref[1].weight.grad, 1e-3, 1e-3)
    close(unit[1].bias.grad, ref[1].bias.grad, 1e-3, 1e-3)


def test_grouped_statistics_two_threads():
    """Two forward_pair-style passes on two Python threads (what nn.DataParallel does with replicas,
    train.py:540-541): batch-statistic groups are call arguments, so the passes cannot disturb each other."""
    import threading

    units = [load_procedural(sm.convbn(32, 32, 3, 1, 1, 1), "t.thr.").to(DEV).train() for _ in range(2)]
    x = seeded((4, 32, 24, 40), 91).to(DEV).contiguous(memory_format=CL)
    with torch.no_grad():
        want1 = sm._convbn_unit(x, load_procedural(sm.convbn(32, 32, 3, 1, 1, 1), "t.thr.").to(DEV).train(), groups=1)
        want2 = sm._convbn_unit(x, load_procedural(sm.convbn(32, 32, 3, 1, 1, 1), "t.thr.").to(DEV).train(), groups=2)
    out = [None, None]

    def work(i, g):
        with torch.no_grad():
            for _ in range(20):
                fresh = load_procedural(sm.convbn(32, 32, 3, 1, 1, 1), "t.thr.").to(DEV).train()
                out[i] = sm._convbn_unit(x, fresh, groups=g)
        torch.cuda.synchronize()

    ts = [threading.Thread(target=work, args=(0, 1)), threading.Thread(target=work, args=(1, 2))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert torch.equal(out[0], want1) and torch.equal(out[1], want2)
    assert not torch.equal(want1, want2)
    del units
