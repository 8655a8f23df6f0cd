"""GPU parity tests: every HIP kernel, called through the C ABI, against the
oracle (oracle/) on the same seeded inputs, and against the committed goldens."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from activezero_amd import ops  # noqa: E402
from activezero_amd.utils import reprojection as rp  # noqa: E402
from activezero_amd.utils import warp_ops  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from oracle import reprojection_oracle as ro  # noqa: E402
from oracle import warp_oracle as wo  # noqa: E402
from tests._weights import seeded  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"
T = torch.from_numpy


def dev(a):
    if isinstance(a, np.ndarray):
        a = T(a)
    return a.to(DEV).contiguous()


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ K1/K2
def test_warp_scatter_known_answers():
    kat = json.load(open(os.path.join(HERE, "golden", "g5_warp_kat.json")))
    for case in kat["cases"]:
        src = np.asarray(case["src"], np.float32)[None, None]
        disp = np.asarray(case["disp"], np.int32)[None, None]
        out = warp_ops.apply_disparity_cu(dev(src), dev(disp))
        assert np.array_equal(out.cpu().numpy()[0, 0], np.asarray(case["expect"], np.float32)), case["name"]


@pytest.mark.parametrize("shape", [(2, 3, 16, 33), (1, 1, 5, 1), (2, 1, 64, 960), (1, 2, 3, 1500)])
@pytest.mark.parametrize("sign", [1, -1])
def test_warp_scatter_vs_oracle_bit_exact(shape, sign):
    rng = np.random.default_rng(sum(shape) + sign)
    n, c, h, w = shape
    src = rng.normal(size=shape).astype(np.float32)
    disp = (sign * rng.integers(0, max(2, w // 3), size=(n, h, w))).astype(np.int32)
    out = warp_ops.apply_disparity_cu(dev(src), dev(disp))
    assert np.array_equal(out.cpu().numpy(), wo.warp_scatter_c(src, disp))
    # [N,1,H,W] disparity and the train.py usage apply_disparity_cu(x, x.int())
    out2 = warp_ops.apply_disparity_cu(dev(src), dev(disp[:, None]))
    assert torch.equal(out, out2)


def test_warp_scatter_train_usage_and_preconditions():
    x = (40 * seeded((2, 1, 32, 96), 7, 0, 1)).to(DEV)
    out = warp_ops.apply_disparity_cu(x, x.type(torch.int))
    exp = wo.apply_disparity_cu_oracle(x.cpu(), x.cpu().int())
    assert torch.equal(out.cpu(), exp)
    with pytest.raises(AssertionError):
        mixed = torch.tensor([[[[1, -1, 0, 0]]]], dtype=torch.int32, device=DEV)
        warp_ops.apply_disparity_cu(torch.zeros(1, 1, 1, 4, device=DEV), mixed)
    with pytest.raises(AssertionError):
        warp_ops.apply_disparity_cu(x, x)  # float disparity
    with pytest.raises(AssertionError):
        warp_ops.apply_disparity_cu(x.cpu(), x.cpu().int())


def test_warp_scatter_full_size_properties():
    # 544x960: zero disparity is the identity; a constant shift is a shifted copy
    x = seeded((4, 1, 544, 960), 11).to(DEV)
    z = torch.zeros(4, 544, 960, dtype=torch.int32, device=DEV)
    assert torch.equal(warp_ops.apply_disparity_cu(x, z), x)
    s = warp_ops.apply_disparity_cu(x, z + 7)
    assert torch.equal(s[..., 7:], x[..., :-7]) and s[..., :7].abs().sum() == 0


# ------------------------------------------------------------------ K3
def test_cost_volume_golden_and_oracle(golden):
    g = golden("g1_cost_volume")
    b, c, h, w, nd = [int(v) for v in g["shape"]]
    fl = seeded((b, c, h, w), 101).to(DEV).requires_grad_()
    fr = seeded((b, c, h, w), 102).to(DEV).requires_grad_()
    vol = ops.cost_volume(fl, fr, nd)
    assert np.array_equal(vol.detach().cpu().numpy(), g["cost"])
    gl, gr = torch.autograd.grad(vol, (fl, fr), seeded(tuple(vol.shape), 103).to(DEV))
    close(gl, g["grad_l"], 1e-5, 1e-5)
    close(gr, g["grad_r"], 1e-5, 1e-5)


@pytest.mark.parametrize("shape,nd", [((1, 32, 5, 7), 3), ((2, 32, 9, 30), 12), ((1, 8, 4, 10), 16),
                                      ((1, 32, 68, 240), 48)])
def test_cost_volume_layouts_vs_oracle(shape, nd):
    b, c, h, w = shape
    fl, fr = seeded(shape, 1), seeded(shape, 2)
    ref = po.build_cost_volume(fl.clone().requires_grad_(), fr.clone().requires_grad_(), nd)
    cot = seeded(tuple(ref.shape), 3)
    a, bb = fl.clone().requires_grad_(), fr.clone().requires_grad_()
    refv = po.build_cost_volume(a, bb, nd)
    rgl, rgr = torch.autograd.grad(refv, (a, bb), cot)
    # NCDHW
    x, y = fl.to(DEV).requires_grad_(), fr.to(DEV).requires_grad_()
    vol = ops.cost_volume(x, y, nd)
    assert torch.equal(vol.detach().cpu(), refv.detach())
    gl, gr = torch.autograd.grad(vol, (x, y), cot.to(DEV))
    close(gl, rgl, 1e-5, 1e-5)
    close(gr, rgr, 1e-5, 1e-5)
    # NDHWC
    xl = fl.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    yl = fr.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    vol_cl = ops.cost_volume_ndhwc(xl, yl, nd)
    assert torch.equal(vol_cl.detach().permute(0, 4, 1, 2, 3).cpu(), refv.detach())
    gl, gr = torch.autograd.grad(vol_cl, (xl, yl), cot.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    close(gl.permute(0, 3, 1, 2), rgl, 1e-5, 1e-5)
    close(gr.permute(0, 3, 1, 2), rgr, 1e-5, 1e-5)


def test_cost_volume_full_size_properties():
    # config 2 shape (B=1 to bound memory): structure checks that do not need the oracle
    b, c, h, w, nd = 1, 32, 136, 240, 48
    fl, fr = seeded((b, c, h, w), 5).to(DEV), seeded((b, c, h, w), 6).to(DEV)
    vol = ops.cost_volume(fl, fr, nd)
    assert vol.shape == (b, 2 * c, nd, h, w)
    for i in (0, 1, 17, 47):
        assert torch.equal(vol[:, :c, i, :, i:], fl[..., i:])
        assert torch.equal(vol[:, c:, i, :, i:], fr[..., : w - i])
        assert vol[:, :, i, :, :i].abs().sum() == 0
    # adjoint identity <vol(fl,fr), G> == <fl, gl> + <fr, gr>
    G = seeded(tuple(vol.shape), 8).to(DEV)
    fl2, fr2 = fl.clone().requires_grad_(), fr.clone().requires_grad_()
    v2 = ops.cost_volume(fl2, fr2, nd)
    gl, gr = torch.autograd.grad(v2, (fl2, fr2), G)
    lhs = (v2.double() * G.double()).sum()
    rhs = (fl.double() * gl.double()).sum() + (fr.double() * gr.double()).sum()
    assert abs(lhs.item() - rhs.item()) <= 1e-6 * abs(lhs.item()) + 1e-3


# ------------------------------------------------------------------ K6
def test_softargmin_golden(golden):
    g = golden("g2_softargmin")
    for k in (1, 2, 3):
        c = dev(g[f"cost{k}"]).requires_grad_()
        p = ops.softargmin(c)
        close(p, g[f"pred{k}"], 1e-5, 2e-5)
        (gr,) = torch.autograd.grad(p, c, dev(g[f"cot{k}"]))
        close(gr, g[f"grad{k}"], 1e-4, 1e-5)
    g = golden("g2_softargmin_d192")
    c = dev(g["cost"]).requires_grad_()
    p = ops.softargmin(c)
    close(p, g["pred"], 1e-5, 1e-4)  # 1e-4 px on a 0..191 range
    (gr,) = torch.autograd.grad(p, c, dev(g["cot"]))
    close(gr, g["grad"], 1e-4, 2e-5)


@pytest.mark.parametrize("shape", [(1, 48, 9, 21), (2, 16, 17, 16), (1, 8, 1, 1), (3, 6, 5, 7),
                                   (1, 48, 34, 60)])
def test_softargmin_vs_oracle(shape):
    b, d, h, w = shape
    lg = 5 * seeded((b, 1, d, h, w), 21)
    a = lg.clone().requires_grad_()
    ref = po.soft_argmin_head(a, 4 * d, 4 * h, 4 * w)
    cot = seeded(tuple(ref.shape), 22)
    (rg,) = torch.autograd.grad(ref, a, cot)
    x = lg.to(DEV).requires_grad_()
    out = ops.softargmin(x)
    close(out, ref, 1e-5, 1e-4)
    (gg,) = torch.autograd.grad(out, x, cot.to(DEV))
    close(gg, rg, 2e-4, 2e-5)


def test_softargmin_full_size_properties():
    # D=192 at 544x960: (i) constant logits -> uniform softmax -> (D-1)/2;
    # (ii) a sharp peak at plane k -> disparity close to the peak's centre 4k+1.5;
    # (iii) shift invariance of softmax.
    b, d, h, w = 2, 48, 136, 240
    x = torch.zeros(b, 1, d, h, w, device=DEV)
    out = ops.softargmin(x)
    assert out.shape == (b, 1, 544, 960)
    assert torch.allclose(out, torch.full_like(out, 95.5), atol=1e-3)
    y = seeded((b, 1, d, h, w), 31).to(DEV)
    o1, o2 = ops.softargmin(y), ops.softargmin(y + 3.25)
    assert torch.allclose(o1, o2, atol=1e-3)
    x[:, :, 20] = 60.0
    pk = ops.softargmin(x)
    assert torch.allclose(pk, torch.full_like(pk, 81.5), atol=1e-2)
    # (iv) a spike so tall that every other upsampled level underflows: the shift must
    # be the max of the UPSAMPLED logits or the sum would be 0 (-> NaN)
    x[:, :, 20] = 4000.0
    pk = ops.softargmin(x)
    assert torch.isfinite(pk).all()
    assert torch.allclose(pk, torch.full_like(pk, 81.5), atol=1e-3)
    assert (o1 >= 0).all() and (o1 <= 191).all()


# ------------------------------------------------------------------ K7
def test_warp_gather_golden(golden):
    g = golden("g6_apply_disparity")
    d = dev(g["disp"]).requires_grad_()
    img = dev(g["img"]).requires_grad_()
    out = rp.apply_disparity(img, d)
    close(out, g["out"], 1e-5, 1e-5)
    gd, gi = torch.autograd.grad(out, (d, img), dev(g["cot"]))
    close(gd, g["grad"], 1e-4, 1e-5)
    # image gradient against the oracle
    a, bimg = T(g["disp"]).requires_grad_(), T(g["img"]).requires_grad_()
    (ri,) = torch.autograd.grad(ro.apply_disparity(bimg, a), bimg, T(g["cot"]))
    close(gi, ri, 1e-4, 1e-5)


@pytest.mark.parametrize("shape", [(1, 1, 16, 24), (2, 4, 33, 65), (1, 121, 12, 20)])
def test_warp_gather_vs_oracle(shape):
    b, c, h, w = shape
    img = seeded(shape, 41)
    disp = seeded((b, 1, h, w), 42, -8.0, 8.0)
    a = disp.clone().requires_grad_()
    ref = ro.apply_disparity(img, a)
    cot = seeded(shape, 43)
    (rg,) = torch.autograd.grad(ref, a, cot)
    d = disp.to(DEV).requires_grad_()
    out = rp.apply_disparity(img.to(DEV), d)
    close(out, ref, 1e-4, 2e-5)
    (gd,) = torch.autograd.grad(out, d, cot.to(DEV))
    close(gd, rg, 1e-3, 1e-4)


# ------------------------------------------------------------------ K8
@pytest.mark.parametrize("kind", ["pat", "con"])
@pytest.mark.parametrize("ps", [1, 3, 11])
@pytest.mark.parametrize("use_mask", [False, True])
def test_patch_reproj_golden(golden, kind, ps, use_mask):
    g = golden("g7_reproj_patch")
    d = dev(g["disp"]).requires_grad_()
    m = dev(g["mask"]) if use_mask else None
    loss, vis, mo = rp.get_reproj_error_patch(dev(g[kind + "_l"]), dev(g[kind + "_r"]), d, m, ps)
    tag = f"{kind}_ps{ps}_{'mask' if use_mask else 'nomask'}"
    close(loss, g[tag + "_loss"], 1e-5, 1e-7)
    close(vis, g[tag + "_vis"], 1e-4, 1e-4)
    assert np.array_equal(mo.cpu().numpy(), g[tag + "_m"])
    loss.backward()
    close(d.grad, g[tag + "_grad"], 1e-3, 1e-7)


def test_patch_reproj_two_channels_and_scaled_grad(golden):
    g = golden("g7_reproj_patch")
    d = dev(g["c2_disp"]).requires_grad_()
    loss, vis, mo = rp.get_reproj_error_patch(dev(g["c2_l"]), dev(g["c2_r"]), d, None, 3)
    close(loss, g["c2_loss"], 1e-5, 1e-7)
    close(vis, g["c2_vis"], 1e-4, 1e-4)
    (3.0 * loss).backward()
    close(d.grad, 3.0 * g["c2_grad"], 1e-3, 1e-7)


def test_patch_reproj_mid_size_vs_oracle():
    b, h, w, ps = 1, 64, 96, 11
    pl = (seeded((b, 1, h, w), 51, 0, 1) < 0.25).float()
    disp = 2.0 + 10.0 * seeded((b, 1, h, w), 52, 0, 1)
    pr = (seeded((b, 1, h, w), 53, 0, 1) < 0.25).float()
    mask = seeded((b, 1, h, w), 54, 0, 1) < 0.8
    a = disp.clone().requires_grad_()
    rl, rv, rm = ro.get_reproj_error_patch(pl, pr, a, mask, ps)
    rl.backward()
    d = disp.to(DEV).requires_grad_()
    loss, vis, mo = rp.get_reproj_error_patch(pl.to(DEV), pr.to(DEV), d, mask.to(DEV), ps)
    close(loss, rl, 1e-5, 1e-7)
    close(vis, rv, 1e-4, 1e-4)
    assert torch.equal(mo.cpu(), rm)
    loss.backward()
    close(d.grad, a.grad, 1e-3, 1e-8)


def test_image_reprojection_variants_golden(golden):
    g = golden("g7_reproj_image")
    d = dev(g["disp"]).requires_grad_()
    lo, wa, mo = rp.get_reprojection_error_old(dev(g["img_l"]), dev(g["img_r"]), d, dev(g["mask"]))
    close(lo, g["old_loss"], 1e-5, 1e-7)
    close(wa, g["old_warped"], 1e-4, 1e-5)
    assert np.array_equal(mo.cpu().numpy(), g["old_mask"])
    (go,) = torch.autograd.grad(lo, d)
    close(go, g["old_grad"], 1e-3, 1e-7)
    tot, stages, parts = rp.get_reprojection_error_diff_ratio(
        dev(g["img_l"]), dev(g["img_r"]), d, dev(g["mask"]))
    close(tot, g["dr_total"], 1e-5, 1e-7)
    close(stages["stage0"]["warped"], g["dr_warped0"], 1e-4, 1e-5)
    close([parts[f"stage{i}"] for i in range(3)], g["dr_parts"], 1e-5, 1e-7)
    (gd,) = torch.autograd.grad(tot, d)
    close(gd, g["dr_grad"], 1e-3, 1e-7)


def test_get_reprojection_error_two_sided_vs_oracle():
    il, ir = seeded((1, 2, 24, 40), 61), seeded((1, 2, 24, 40), 62)
    dl, dr = seeded((1, 1, 24, 40), 63, 0.5, 9.0), seeded((1, 1, 24, 40), 64, 0.5, 9.0)
    ref = ro.get_reprojection_error(il, ir, dl, dr)
    out = rp.get_reprojection_error(il.to(DEV), ir.to(DEV), dl.to(DEV), dr.to(DEV))
    close(out[0], ref[0], 1e-5, 1e-7)
    close(out[1], ref[1], 1e-5, 1e-7)
    close(out[2], ref[2], 1e-4, 1e-5)
    assert torch.equal(out[4].cpu(), ref[4]) and torch.equal(out[5].cpu(), ref[5])


# ------------------------------------------------------------------ K9
def test_lcn_golden(golden):
    g = golden("g8_lcn")
    for k in (3, 9):
        n, s = rp.local_contrast_norm(dev(g["img"]), k)
        close(n, g[f"k{k}_normed"], 1e-4, 1e-4)
        close(s, g[f"k{k}_std"], 1e-5, 1e-6)
    n, s = rp.local_contrast_norm(dev(g["img3"]), 5)
    close(n, g["c3_normed"], 1e-4, 1e-4)
    close(s, g["c3_std"], 1e-5, 1e-6)


def test_lcn_full_size_flat_and_oracle():
    img = seeded((1, 1, 540, 960), 71, 0, 1)
    img[:, :, 100:200, 100:300] = 0.5  # flat region: std must be exactly 0 inside
    n, s = rp.local_contrast_norm(img.to(DEV), 11)
    rn, rs = ro.local_contrast_norm(img, 11)
    close(s, rs, 1e-4, 1e-6)
    close(n, rn, 1e-3, 1e-3)
    assert s[0, 0, 110:190, 110:290].abs().max().item() == 0.0


def test_dispatcher_ops_match_the_python_wrappers():
    """torch.ops.azhip.* (activezero_amd/torch_ops.py) against activezero_amd.ops on the same inputs,
    values and gradients"""
    import activezero_amd.torch_ops  # noqa: F401

    lg = (3 * seeded((2, 6, 5, 7), 901)).to(DEV)
    a, b = lg.clone().requires_grad_(), lg.clone().requires_grad_()
    pa, pb = torch.ops.azhip.softargmin(a), ops.softargmin(b)
    assert torch.equal(pa, pb)
    ct = seeded(tuple(pa.shape), 902).to(DEV)
    pa.backward(ct); pb.backward(ct)
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-5)  # (float atomics: last-bit run-to-run differences)
    fl, fr = seeded((2, 32, 6, 20), 903).to(DEV), seeded((2, 32, 6, 20), 904).to(DEV)
    x, y = fl.clone().requires_grad_(), fr.clone().requires_grad_()
    v = torch.ops.azhip.cost_volume(x, y, 5)
    assert torch.equal(v, ops.cost_volume(fl, fr, 5))
    v.backward(torch.ones_like(v))
    x2, y2 = fl.clone().requires_grad_(), fr.clone().requires_grad_()
    ops.cost_volume(x2, y2, 5).backward(torch.ones_like(v))
    assert torch.equal(x.grad, x2.grad) and torch.equal(y.grad, y2.grad)
    img, dsp = seeded((2, 3, 9, 17), 905).to(DEV), seeded((2, 1, 9, 17), 906, -3, 3).to(DEV)
    d1, d2 = dsp.clone().requires_grad_(), dsp.clone().requires_grad_()
    w1, w2 = torch.ops.azhip.warp_gather(img, d1), ops.warp_gather(img, d2)
    assert torch.equal(w1, w2)
    w1.sum().backward(); w2.sum().backward()
    assert torch.equal(d1.grad, d2.grad)


@pytest.mark.parametrize("mode,cin,cout", [(0, 32, 32), (1, 32, 64), (2, 64, 32)])
def test_dispatcher_conv3d_bn3d_patch_reproj(mode, cin, cout):
    """torch.ops.azhip.conv3d / bn3d / patch_reproj: values and registered gradients against torch's own operators in
    fp64 (conv) and against the Python wrappers (patch reprojection)"""
    import torch.nn.functional as F
    import activezero_amd.torch_ops  # noqa: F401

    b, d, h, w = 2, 4, 6, 16
    x = seeded((b, d, h, w, cin), 911).to(DEV).requires_grad_()
    wshape = (cin, cout, 3, 3, 3) if mode == 2 else (cout, cin, 3, 3, 3)
    wt = (0.1 * seeded(wshape, 912)).to(DEV).requires_grad_()
    y = torch.ops.azhip.conv3d(x, wt, mode)
    xd = x.detach().double().permute(0, 4, 1, 2, 3).requires_grad_()
    wd = wt.detach().double().requires_grad_()
    if mode == 2:
        yd = F.conv_transpose3d(xd, wd, stride=2, padding=1, output_padding=1)
    else:
        yd = F.conv3d(xd, wd, stride=1 + mode, padding=1)
    ydl = yd.permute(0, 2, 3, 4, 1)
    assert y.shape == ydl.shape
    sc_ = float(ydl.abs().max())
    assert float((y.double() - ydl).abs().max()) <= 3e-6 * sc_
    # bn3d on top (affine + residual + ReLU) and a backward through both
    scale, shift = (1.0 + 0.1 * seeded((cout,), 913)).to(DEV).requires_grad_(), (0.1 * seeded((cout,), 914)).to(DEV).requires_grad_()
    res = seeded(tuple(y.shape), 915).to(DEV).requires_grad_()
    z = torch.ops.azhip.bn3d(y, scale, shift, res, True)
    zd = F.relu(ydl * scale.detach().double() + shift.detach().double() + res.detach().double())
    assert float((z.double() - zd).abs().max()) <= 1e-5 * max(1.0, float(zd.abs().max()))
    ct = seeded(tuple(z.shape), 916).to(DEV)
    z.backward(ct)
    (zd * ct.double()).sum().backward()
    gx_ref = xd.grad.permute(0, 2, 3, 4, 1)
    assert float((x.grad.double() - gx_ref).abs().max()) <= 1e-5 * float(gx_ref.abs().max())
    assert float((wt.grad.double() - wd.grad).abs().max()) <= 1e-5 * float(wd.grad.abs().max())
    if mode == 0:  # patch reprojection once
        pl = (seeded((1, 1, 32, 48), 917) > 0.4).float().to(DEV)
        pr = pl.roll(2, 3).contiguous()
        dsp = (3.0 * seeded((1, 1, 32, 48), 918).abs()).to(DEV)
        d1, d2 = dsp.clone().requires_grad_(), dsp.clone().requires_grad_()
        l1 = torch.ops.azhip.patch_reproj(pl, pr, d1, None, 5)
        l2 = ops.patch_reprojection(pl, pr, d2, None, 5, want_vis=False)[0]
        assert torch.equal(l1, l2)
        l1.backward(); l2.backward()
        assert torch.allclose(d1.grad, d2.grad, rtol=1e-5, atol=1e-7)
