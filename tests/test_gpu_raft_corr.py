"""GPU parity: RAFT-Stereo 1-D correlation volume / pyramid / lookup (K10/K11) against the
golden produced by the imported reference CorrBlock1D and against plain torch ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd.nets.raft.corr import CorrBlock1D  # noqa: E402
from tests._weights import seeded  # noqa: E402

DEV = "cuda:0"
T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_corr1d_golden(golden):
    g = golden("g9_corr1d")
    f1, f2 = T(g["fmap1"]).to(DEV).requires_grad_(), T(g["fmap2"]).to(DEV).requires_grad_()
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    assert len(blk.corr_pyramid) == 5
    for i in range(4):
        assert tuple(blk.corr_pyramid[i].shape) == tuple(g[f"pyr{i}"].shape)
        close(blk.corr_pyramid[i], g[f"pyr{i}"], 1e-5, 2e-6)
    out = blk(T(g["coords"]).to(DEV))
    close(out, g["out"], 1e-4, 2e-5)
    g1, g2 = torch.autograd.grad(out, (f1, f2), T(g["cot"]).to(DEV))
    close(g1, g["grad1"], 1e-4, 2e-5)
    close(g2, g["grad2"], 1e-4, 2e-5)


def _torch_reference(f1, f2, coords, levels=4, r=4):
    b, c, h, w1 = f1.shape
    corr = torch.einsum("aijk,aijh->ajkh", f1, f2) / torch.sqrt(torch.tensor(float(c)))
    pyr = [corr.reshape(b * h * w1, 1, 1, -1)]
    for _ in range(levels):
        pyr.append(F.avg_pool2d(pyr[-1], [1, 2], stride=[1, 2]))
    cx = coords[:, :1].permute(0, 2, 3, 1).reshape(b * h * w1, 1, 1, 1)
    outs = []
    for i in range(levels):
        dx = torch.linspace(-r, r, 2 * r + 1).view(1, 1, 2 * r + 1, 1)
        x0 = dx + cx / 2 ** i
        wl = pyr[i].shape[-1]
        grid = torch.cat([2 * x0 / (wl - 1) - 1, torch.zeros_like(x0)], -1)
        outs.append(F.grid_sample(pyr[i], grid, align_corners=True).view(b, h, w1, -1))
    return torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous(), pyr


@pytest.mark.parametrize("shape", [(1, 256, 5, 60), (2, 64, 3, 75), (1, 32, 2, 240)])
def test_corr1d_vs_torch_ops(shape):
    b, c, h, w = shape
    f1, f2 = seeded(shape, 1), seeded(shape, 2)
    coords = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")[::-1], 0).float()[None]
    coords = coords.repeat(b, 1, 1, 1).clone()
    coords[:, 0] -= seeded((b, h, w), 3, -3.0, 0.6 * w)  # includes far out-of-range lookups
    a1, a2 = f1.clone().requires_grad_(), f2.clone().requires_grad_()
    ref, pyr = _torch_reference(a1, a2, coords)
    cot = seeded(tuple(ref.shape), 4)
    r1, r2 = torch.autograd.grad(ref, (a1, a2), cot)
    x1, x2 = f1.to(DEV).requires_grad_(), f2.to(DEV).requires_grad_()
    blk = CorrBlock1D(x1, x2)
    close(blk.corr_pyramid[0], pyr[0], 1e-4, 1e-4)
    close(blk.corr_pyramid[3], pyr[3], 1e-4, 1e-4)
    out = blk(coords.to(DEV))
    close(out, ref, 1e-4, 2e-4)
    g1, g2 = torch.autograd.grad(out, (x1, x2), cot.to(DEV))
    close(g1, r1, 1e-3, 2e-4)
    close(g2, r2, 1e-3, 2e-4)


def test_corr1d_config5_shape_properties():
    # BASELINE config 5 feature shape: symmetric inputs -> symmetric volume; lookup at the
    # pixel's own column returns the diagonal (the squared norm / sqrt(C))
    b, c, h, w = 1, 256, 136, 240
    f = seeded((b, c, h, w), 7).to(DEV)
    blk = CorrBlock1D(f, f)
    vol = blk.corr_pyramid[0].view(b, h, w, w)
    assert torch.allclose(vol, vol.transpose(2, 3), atol=1e-4)
    coords = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")[::-1], 0).float()[None].to(DEV)
    out = blk(coords)
    assert out.shape == (b, 36, h, w)
    diag = (f * f).sum(1) / 16.0
    assert torch.allclose(out[:, 4], diag, rtol=1e-4, atol=1e-3)


# ---- round 5: the lookups of a step accumulate their pyramid gradients in one buffer per level (az_corr1d_lookup_bwd_acc) -------
@pytest.mark.parametrize("n_lookups", [1, 3, 5])
def test_several_lookups_of_one_pyramid_sum_their_gradients_in_the_kernel(n_lookups, monkeypatch):
    """raft_stereo.py:138-172 looks the same pyramid up once per update; the gradient of fmap1 / fmap2 is the sum over the
    lookups.  With the accumulating kernel only the first lookup node to run hands a buffer to the engine: against the torch
    reference, against the one-buffer-per-lookup route (AZ_LOOKUP_ACC=0), with one lookup's output left out of the loss (its
    node never runs), and over a second backward pass through the retained graph (fresh buffers per pass)."""
    from activezero_amd.nets.raft import corr as C
    b, c, h, w = 2, 64, 3, 75
    f1, f2 = seeded((b, c, h, w), 1), seeded((b, c, h, w), 2)
    base = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")[::-1], 0).float()[None].repeat(b, 1, 1, 1)
    coords = []
    for k in range(n_lookups + 1):
        ck = base.clone()
        ck[:, 0] -= seeded((b, h, w), 10 + k, -3.0, 0.6 * w)
        coords.append(ck)
    cots = [seeded((b, 36, h, w), 30 + k) for k in range(n_lookups + 1)]
    a1, a2 = f1.clone().requires_grad_(), f2.clone().requires_grad_()
    ref = sum((_torch_reference(a1, a2, coords[k])[0] * cots[k]).sum() for k in range(n_lookups))
    r1, r2 = torch.autograd.grad(ref, (a1, a2))
    res = {}
    for acc in (True, False):
        monkeypatch.setattr(C, "LOOKUP_ACC", acc)
        x1, x2 = f1.to(DEV).requires_grad_(), f2.to(DEV).requires_grad_()
        blk = CorrBlock1D(x1, x2)
        outs = [blk(coords[k].to(DEV)) for k in range(n_lookups + 1)]  # the last one stays out of the loss
        loss = sum((outs[k] * cots[k].to(DEV)).sum() for k in range(n_lookups))
        g1, g2 = torch.autograd.grad(loss, (x1, x2), retain_graph=True)
        close(g1, r1, 1e-3, 3e-4 * n_lookups)
        close(g2, r2, 1e-3, 3e-4 * n_lookups)
        h1, h2 = torch.autograd.grad(loss, (x1, x2))  # a second pass: not twice the sums
        torch.testing.assert_close(h1, g1, rtol=1e-5, atol=1e-6 * float(g1.abs().max()))
        torch.testing.assert_close(h2, g2, rtol=1e-5, atol=1e-6 * float(g2.abs().max()))
        res[acc] = (g1, g2)
    for a_, b_ in zip(res[True], res[False]):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-6 * float(b_.abs().max()))
