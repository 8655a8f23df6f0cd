"""GPU parity of K12 (fused disparity loss + error metrics) through the C ABI: against the
reference's outputs (golden G10) and against the oracle at the benchmark's full map size."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from activezero_amd.utils import cascade_metrics as cm  # noqa: E402
from activezero_amd.utils import disp_losses  # noqa: E402
from oracle import metrics_oracle as mo  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import seeded  # noqa: E402

DEV = "cuda:0"
T = torch.from_numpy


def dev(a):
    if isinstance(a, np.ndarray):
        a = T(a)
    return a.to(DEV).contiguous()


def close(a, b, rtol=1e-6, atol=1e-9):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_loss_golden(golden):
    g = golden("g10_metrics")
    gt, mask = dev(g["gt"]), dev(g["mask"])
    preds = [dev(g[k]).requires_grad_() for k in ("pred3", "pred2", "pred1")]
    loss = disp_losses.psmnet_disp(tuple(preds), gt, mask)
    close(loss, g["loss"], rtol=2e-6)
    grads = torch.autograd.grad(loss * 1.0, preds)
    for got, k in zip(grads, ("grad3", "grad2", "grad1")):
        close(got, g[k], rtol=2e-6, atol=1e-10)
    # the in-kernel range rule (train.py:272) selects the same pixels as the explicit mask
    preds2 = [p.detach().clone().requires_grad_() for p in preds]
    loss2 = disp_losses.psmnet_disp_range(tuple(preds2), gt, float(g["maxdisp"]))
    assert loss2.item() == loss.item()
    for a, b in zip(torch.autograd.grad(loss2, preds2), grads):
        assert torch.equal(a, b)
    # upstream gradient is honoured
    preds3 = [p.detach().clone().requires_grad_() for p in preds]
    (disp_losses.psmnet_disp(tuple(preds3), gt, mask) * 2.5).backward()
    close(preds3[0].grad, 2.5 * g["grad3"], rtol=2e-6, atol=1e-10)


def test_loss_empty_mask_is_nan_like_the_reference():
    gt = torch.zeros(1, 1, 4, 8, device=DEV)
    preds = tuple(torch.ones(1, 1, 4, 8, device=DEV) for _ in range(3))
    assert math.isnan(disp_losses.psmnet_disp(preds, gt, gt > 0).item())  # mean over nothing


def test_loss_rejects_bad_inputs():
    gt = torch.zeros(1, 1, 4, 8, device=DEV)
    p = torch.ones(1, 1, 4, 8, device=DEV)
    with pytest.raises(RuntimeError):
        disp_losses.psmnet_disp((p, p, p[..., :4]), gt, gt > 0)
    with pytest.raises(RuntimeError):
        disp_losses.psmnet_disp((p, p, p), gt, (gt > 0)[..., :4])
    with pytest.raises(RuntimeError):
        disp_losses.psmnet_disp((p.cpu(), p, p), gt, gt > 0)


def test_metrics_golden(golden):
    g = golden("g10_metrics")
    gt, mask, zg, dp = (dev(g[k]) for k in ("gt", "mask", "depth_gt", "disp_pred"))
    focal, base = dev(g["focal"]), dev(g["baseline"])
    keys = [str(k) for k in g["metric_keys"]]
    m = cm.compute_err_metric(gt, zg, dp, focal, base, mask)
    assert sorted(m) == keys
    close([m[k] for k in keys], g["metrics"], rtol=2e-6)
    m = cm.compute_err_metric(gt, zg, dp, focal, base, mask, depth_pred=dev(g["depth_pred"]))
    close([m[k] for k in keys], g["metrics_dp"], rtol=2e-6)
    obj = cm.compute_obj_err(gt[:1], zg[:1], dp[:1], focal[:1], base[:1], dev(g["label"]), mask[:1])
    for i, o in enumerate(obj):
        close(o, g[f"obj{i}"], rtol=2e-6)


def test_full_size_against_oracle():
    """BASELINE configs[1] map size (B=4, 544x960): loss, gradients and metrics vs the oracle."""
    b, h, w, md = 4, 544, 960, 192.0
    gt = seeded((b, 1, h, w), 2001, -5.0, 200.0)
    mask = po.disparity_mask(gt, md)
    preds = [(gt + seeded((b, 1, h, w), 2002 + k, -2.5, 2.5)) for k in range(3)]
    ref_p = [p.clone().requires_grad_() for p in preds]
    ref = po.psmnet_disp_loss(tuple(ref_p), gt, mask)
    ref_g = torch.autograd.grad(ref, ref_p)
    got_p = [dev(p).requires_grad_() for p in preds]
    got = disp_losses.psmnet_disp_range(tuple(got_p), dev(gt), md)
    close(got, ref, rtol=5e-6)  # fp64 accumulation here, fp32 pairwise sums in the oracle
    for a, r in zip(torch.autograd.grad(got, got_p), ref_g):
        close(a, r, rtol=5e-6, atol=1e-12)
    focal = torch.full((b, 1, 1, 1), 450.0)
    base = torch.full((b, 1, 1, 1), 0.055)
    zg = torch.where(gt > 0, focal * base / gt.clamp(min=1e-3), torch.zeros_like(gt))
    dp = preds[0].clamp(min=0.5)
    want = mo.compute_err_metric(gt, zg, dp, focal, base, mask)
    have = cm.compute_err_metric(dev(gt), dev(zg), dev(dp), dev(focal), dev(base), dev(mask))
    for k in want:
        close(have[k], want[k], rtol=5e-6)
