"""CPU: the oracle (oracle/) against the golden vectors produced by the imported
reference (tools/make_goldens.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import psmnet_oracle as po
from oracle import reprojection_oracle as ro
from oracle import warp_oracle as wo
from tests._weights import load_bn_buffers, load_procedural, seeded

T = torch.from_numpy
HERE = os.path.dirname(os.path.abspath(__file__))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_g1_cost_volume(golden):
    g = golden("g1_cost_volume")
    b, c, h, w, nd = g["shape"]
    fl = seeded((b, c, h, w), 101).requires_grad_()
    fr = seeded((b, c, h, w), 102).requires_grad_()
    vol = po.build_cost_volume(fl, fr, int(nd))
    assert np.array_equal(vol.detach().numpy(), g["cost"])  # pure copies: bit exact
    gl, gr = torch.autograd.grad(vol, (fl, fr), seeded(tuple(vol.shape), 103))
    close(gl, g["grad_l"])
    close(gr, g["grad_r"])


def test_g2_softargmin(golden):
    g = golden("g2_softargmin")
    md = int(g["maxdisp"])
    for k in (1, 2, 3):
        c = T(g[f"cost{k}"]).requires_grad_()
        p = po.soft_argmin_head(c, md, 4 * c.shape[3], 4 * c.shape[4])
        close(p, g[f"pred{k}"], 1e-5, 1e-5)
        (gr,) = torch.autograd.grad(p, c, T(g[f"cot{k}"]))
        close(gr, g[f"grad{k}"], 1e-4, 1e-6)
    g = golden("g2_softargmin_d192")
    c = T(g["cost"]).requires_grad_()
    p = po.soft_argmin_head(c, 192, 12, 16)
    close(p, g["pred"], 1e-5, 1e-4)
    (gr,) = torch.autograd.grad(p, c, T(g["cot"]))
    close(gr, g["grad"], 1e-4, 1e-5)


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("skips", [False, True])
def test_g3_hourglass(golden, mode, skips):
    g = golden("g3_hourglass")
    hg = load_procedural(po.HourglassOracle(32), "g3.hg.")
    hg.train(mode == "train")
    x = seeded((1, 32, 8, 8, 12), 301).requires_grad_()
    pre, post = seeded((1, 64, 4, 4, 6), 302), seeded((1, 64, 4, 4, 6), 303)
    o, p, q = hg(x, pre if skips else None, post if skips else None)
    tag = f"{mode}_{'skip' if skips else 'noskip'}"
    close(o, g[tag + "_out"], 1e-4, 1e-5)
    close(p, g[tag + "_pre"], 1e-4, 1e-5)
    close(q, g[tag + "_post"], 1e-4, 1e-5)
    ((o * seeded((1, 32, 8, 8, 12), 304)).sum() + p.sum() * 0.25 + q.sum() * 0.5).backward()
    close(x.grad, g[tag + "_gx"], 1e-4, 1e-5)
    close(hg.conv5[0].weight.grad[:8, :8], g[tag + "_gw_conv5"], 1e-4, 1e-4)
    close(hg.conv2[1].weight.grad, g[tag + "_ggamma_conv2"], 1e-4, 1e-4)


@pytest.mark.parametrize("variant,nin", [("psmnet3", 3), ("psmnet6", 6)])
def test_g4_full_psmnet(golden, variant, nin):
    g = golden("g4_" + variant)
    md = int(g["maxdisp"])
    model = load_bn_buffers(load_procedural(po.PSMNetOracle(md, in_ch=nin), "g4."), g)
    assert sorted(model.state_dict().keys()) == list(g["keys"])  # 514 reference keys
    assert len(model.state_dict()) == int(g["nkeys"])
    imgs = [seeded((2, 3, 256, 256), 400 + i, -2.0, 2.0) for i in range(4)]
    args = imgs[:2] if nin == 3 else [imgs[0], imgs[1], imgs[2], imgs[3]]
    st = int(g["pred_stride"])
    model.eval()
    with torch.no_grad():
        pe = model(*args)
    close(pe[..., ::st, ::st], g["pred_eval"], 1e-4, 1e-4)
    model.train()
    preds = model(*args)
    for p, k in zip(preds, ("pred3", "pred2", "pred1")):
        close(p[..., ::st, ::st], g[k], 1e-4, 1e-4)
    gt = T(g["gt"])
    loss = po.psmnet_disp_loss(preds, gt, po.disparity_mask(gt, md))
    close(loss, g["loss"], 1e-5, 1e-6)
    loss.backward()
    sd = dict(model.named_parameters())
    close(sd["classif3.2.weight"].grad, g["g_classif3_2"], 1e-3, 1e-5)
    close(sd["dres0.0.0.weight"].grad[:4, :4], g["g_dres0_0_0"], 1e-3, 1e-5)
    close(dict(model.named_buffers())["dres0.0.1.running_var"], g["rv_dres0"], 1e-5, 1e-6)


def test_g5_warp_known_answers():
    kat = json.load(open(os.path.join(HERE, "golden", "g5_warp_kat.json")))
    for case in kat["cases"]:
        src = np.asarray(case["src"], np.float32)[None, None]
        disp = np.asarray(case["disp"], np.int32)[None]
        exp = np.asarray(case["expect"], np.float32)[None, None]
        assert np.array_equal(wo.warp_scatter_numpy(src, disp), exp), case["name"]
        assert np.array_equal(wo.warp_scatter_c(src, disp), exp), case["name"]
        src3 = np.concatenate([src, 2 * src, -src], 1)  # channels share the disparity plane
        assert np.array_equal(wo.warp_scatter_c(src3, disp), np.concatenate([exp, 2 * exp, -exp], 1))


def test_g5_warp_c_equals_numpy_random():
    rng = np.random.default_rng(5)
    src = rng.normal(size=(2, 3, 16, 33)).astype(np.float32)
    for sign in (1, -1):
        disp = (sign * rng.integers(0, 40, size=(2, 16, 33))).astype(np.int32)
        assert np.array_equal(wo.warp_scatter_c(src, disp), wo.warp_scatter_numpy(src, disp))
    with pytest.raises(AssertionError):
        wo.warp_scatter_c(src, rng.integers(-3, 4, size=(2, 16, 33)).astype(np.int32))


def test_g6_apply_disparity(golden):
    g = golden("g6_apply_disparity")
    d = T(g["disp"]).requires_grad_()
    out = ro.apply_disparity(T(g["img"]), d)
    close(out, g["out"])
    (gr,) = torch.autograd.grad(out, d, T(g["cot"]))
    close(gr, g["grad"], 1e-5, 1e-5)
    # the closed form of the sampled coordinates (what the HIP kernel implements)
    px, py = ro.sample_coords(9, 17, T(g["disp"]))
    j = torch.arange(17.0).view(1, 1, 17)
    assert torch.allclose(px, (j * 17 / 16 + T(g["disp"])[:, 0] - 0.5).double(), atol=1e-5)


@pytest.mark.parametrize("kind", ["pat", "con"])
@pytest.mark.parametrize("ps", [1, 3, 11])
@pytest.mark.parametrize("use_mask", [False, True])
def test_g7_patch(golden, kind, ps, use_mask):
    g = golden("g7_reproj_patch")
    d = T(g["disp"]).requires_grad_()
    m = T(g["mask"]) if use_mask else None
    loss, vis, mo = ro.get_reproj_error_patch(T(g[kind + "_l"]), T(g[kind + "_r"]), d, m, ps)
    tag = f"{kind}_ps{ps}_{'mask' if use_mask else 'nomask'}"
    close(loss, g[tag + "_loss"])
    close(vis, g[tag + "_vis"], 1e-5, 1e-5)
    assert np.array_equal(mo.numpy(), g[tag + "_m"])
    loss.backward()
    close(d.grad, g[tag + "_grad"], 1e-4, 1e-7)


def test_g7_image_variants(golden):
    g = golden("g7_reproj_image")
    d = T(g["disp"]).requires_grad_()
    lo, wa, mo = ro.get_reprojection_error_old(T(g["img_l"]), T(g["img_r"]), d, T(g["mask"]))
    close(lo, g["old_loss"])
    close(wa, g["old_warped"])
    assert np.array_equal(mo.numpy(), g["old_mask"])
    tot, stages, parts = ro.get_reprojection_error_diff_ratio(
        T(g["img_l"]), T(g["img_r"]), d, T(g["mask"]))
    close(tot, g["dr_total"])
    close(stages["stage0"]["warped"], g["dr_warped0"])
    close([parts[f"stage{i}"] for i in range(3)], g["dr_parts"])
    (gd,) = torch.autograd.grad(tot, d)
    close(gd, g["dr_grad"], 1e-4, 1e-7)


def test_g8_lcn(golden):
    g = golden("g8_lcn")
    for k in (3, 9):
        n, s = ro.local_contrast_norm(T(g["img"]), k)
        close(n, g[f"k{k}_normed"], 1e-5, 1e-5)
        close(s, g[f"k{k}_std"], 1e-5, 1e-6)
    n, s = ro.local_contrast_norm(T(g["img3"]), 5)
    close(n, g["c3_normed"], 1e-5, 1e-5)


def test_psmnet_disp_loss_formula():
    gt = 1 + 30 * torch.rand(1, 1, 8, 8)
    ps = [gt + torch.randn_like(gt) for _ in range(3)]
    mask = po.disparity_mask(gt, 16)
    ref = sum(w * F.smooth_l1_loss(p[mask], gt[mask]) for w, p in zip((1.0, 0.7, 0.5), ps))
    assert torch.allclose(po.psmnet_disp_loss(ps, gt, mask), ref)


def test_g10_loss_and_metrics(golden):
    """Loss restatement and error metrics against the reference's outputs (G10)."""
    from oracle import metrics_oracle as mo

    g = golden("g10_metrics")
    gt, mask = T(g["gt"]), T(g["mask"])
    preds = [T(g[k]).requires_grad_() for k in ("pred3", "pred2", "pred1")]
    loss = po.psmnet_disp_loss(tuple(preds), gt, mask)
    close(loss, g["loss"], rtol=1e-6)
    for p, k in zip(preds, ("grad3", "grad2", "grad1")):
        close(torch.autograd.grad(loss, p, retain_graph=True)[0], g[k], rtol=1e-6, atol=1e-9)
    assert torch.equal(mask, po.disparity_mask(gt, float(g["maxdisp"])))
    focal, base, zg, dp = (T(g[k]) for k in ("focal", "baseline", "depth_gt", "disp_pred"))
    keys = [str(k) for k in g["metric_keys"]]
    m = mo.compute_err_metric(gt, zg, dp, focal, base, mask)
    close([m[k] for k in keys], g["metrics"], rtol=1e-6)
    m = mo.compute_err_metric(gt, zg, dp, focal, base, mask, depth_pred=T(g["depth_pred"]))
    close([m[k] for k in keys], g["metrics_dp"], rtol=1e-6)
    obj = mo.compute_obj_err(gt[:1], zg[:1], dp[:1], focal[:1], base[:1], T(g["label"]), mask[:1])
    for i, o in enumerate(obj):
        close(o, g[f"obj{i}"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_g12_convgru(golden, tag):
    """oracle/raft_gru_oracle.py against the imported reference class (nets/raft/update.py:19-41)"""
    from oracle.raft_gru_oracle import ConvGRUOracle
    g = golden("g12_convgru")
    meta = [int(v) for v in g[f"{tag}_meta"]]
    hidden, h, w, b, sd, cs, ps = meta[:7]
    cx = meta[7:]
    mod = load_procedural(ConvGRUOracle(hidden, sum(cx)), f"g12{tag}.")
    hid = torch.tanh(seeded((b, hidden, h, w), sd, -2.0, 2.0))
    ctx = [seeded((b, hidden, h, w), sd + 1 + i, -0.8, 0.8) for i in range(3)]
    xs = [seeded((b, n, h, w), sd + 4 + i, -1.7, 1.7) for i, n in enumerate(cx)]
    lat = lambda t: t[:, ::cs, ::ps, ::ps]
    hr = hid.clone().requires_grad_(tag == "d")
    out = mod(hr, *ctx, *xs)
    close(lat(out), g[f"{tag}_out32"], 1e-5, 1e-6)
    with torch.no_grad():
        hh = hid
        for _ in range(4):
            hh = mod(hh, *ctx, *xs)
        close(lat(hh), g[f"{tag}_iter4_32"], 1e-5, 2e-6)
        mod.double()
        close(lat(mod(hid.double(), *[c.double() for c in ctx], *[x.double() for x in xs])), g[f"{tag}_out64"], 1e-12, 1e-13)
        mod.float()
    if tag == "d":
        (out * seeded((b, hidden, h, w), sd + 9)).sum().backward()
        close(hr.grad, g["d_gh"], 1e-4, 1e-6)
        for name in ("convz", "convr", "convq"):
            close(getattr(mod, name).weight.grad[:16, :32], g[f"d_gw_{name}"], 1e-4, 2e-5)
            close(getattr(mod, name).bias.grad, g[f"d_gb_{name}"], 1e-4, 2e-5)
