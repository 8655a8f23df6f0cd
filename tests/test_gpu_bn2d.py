"""GPU parity of the extractor's BatchNorm2d(+residual)(+ReLU) on the HIP BatchNorm kernels
(activezero_amd/bn2d.py: az_bn3d_stats / finalize / apply / bwd) against torch's own modules,
including the per-group statistics that make one pass over the stacked (left, right) batch equal
to the reference's two sequential calls (nets/psmnet/psmnet_3.py:145-146)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import bn2d  # noqa: E402

DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("has_res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("C", [32, 64, 128])
def test_bn_act_train_matches_torch(C, relu, has_res, groups):
    torch.manual_seed(C + 2 * relu + has_res + 10 * groups)
    n, h, w = 4, 17, 23  # odd sizes: partial tiles in the statistics kernel
    x = (torch.randn(n, C, h, w, device=DEV) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
    res = torch.randn(n, C, h, w, device=DEV).contiguous(memory_format=torch.channels_last) if has_res else None
    gy = torch.randn(n, C, h, w, device=DEV)
    ours, ref = torch.nn.BatchNorm2d(C).to(DEV).train(), torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        ours.weight.uniform_(0.5, 1.5)
        ours.bias.uniform_(-0.5, 0.5)
        ref.load_state_dict(ours.state_dict())
    xa = x.clone().requires_grad_()
    ra = res.clone().requires_grad_() if has_res else None
    ya = bn2d.bn_act(xa, ours, relu, ra, groups)
    ya.backward(gy)
    xb = x.clone().requires_grad_()
    rb = res.clone().requires_grad_() if has_res else None
    parts = []
    for g in range(groups):  # the reference order: one module call per image set
        sl = slice(g * n // groups, (g + 1) * n // groups)
        t = ref(xb[sl])
        if has_res:
            t = t + rb[sl]
        parts.append(F.relu(t) if relu else t)
    yb = torch.cat(parts, 0)
    yb.backward(gy)
    tol = 5e-6
    assert _rel(ya, yb) < tol
    assert _rel(xa.grad, xb.grad) < tol
    assert _rel(ours.weight.grad, ref.weight.grad) < tol and _rel(ours.bias.grad, ref.bias.grad) < tol
    assert _rel(ours.running_mean, ref.running_mean) < tol and _rel(ours.running_var, ref.running_var) < tol
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == groups
    if has_res:
        assert _rel(ra.grad, rb.grad) < tol


def test_bn_act_eval_uses_running_statistics():
    torch.manual_seed(1)
    bn = torch.nn.BatchNorm2d(64).to(DEV)
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    bn.eval()
    x = torch.randn(2, 64, 9, 11, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = bn2d.bn_act(x, bn, relu=True, groups=2)
        want = F.relu(bn(x))
    assert _rel(got, want) < 5e-6


def test_bn_act_rejects_bad_grouping():
    bn = torch.nn.BatchNorm2d(32).to(DEV).train()
    x = torch.randn(3, 32, 8, 8, device=DEV)
    with pytest.raises(RuntimeError):
        bn2d.bn_act(x, bn, groups=2)
