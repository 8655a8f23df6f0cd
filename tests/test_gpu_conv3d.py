"""GPU parity: the MFMA conv3d / deconv3d / BatchNorm3d kernels (K4/K5) against the
oracle's op set (torch CPU conv3d / conv_transpose3d / batch_norm) and the goldens
produced by the imported reference (g3_*)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from activezero_amd import agg3d, conv3d, ops  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_3  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import load_procedural, seeded  # noqa: E402

DEV = "cuda:0"


def cl(x):  # NCDHW (cpu) -> channels-last on the GPU
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def ncdhw(x):  # channels-last (gpu) -> NCDHW cpu
    return x.detach().permute(0, 4, 1, 2, 3).contiguous().cpu()


@pytest.fixture(params=["bf16x6", "fp32", "f16x3"])
def arith(request):
    """every MFMA test of this file runs in both arithmetic modes (passed per call: no global switch)"""
    return conv3d.Arith.of(request.param)


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("dims", [(1, 5, 7, 19), (2, 4, 8, 16), (1, 3, 9, 33), (1, 1, 9, 20), (1, 2, 16, 35)])
def test_conv_stride1_vs_torch(cin, cout, dims, arith):
    b, d, h, w = dims
    x = seeded((b, cin, d, h, w), 1)
    wt = seeded((cout, cin, 3, 3, 3), 2, -0.2, 0.2)
    ref = F.conv3d(x, wt, padding=1)
    out = conv3d.conv_plain(cl(x), wt.to(DEV), conv3d.CONV_S1, arith)
    close(ncdhw(out), ref, 1e-4, 2e-5)


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (32, 32), (64, 32)])
@pytest.mark.parametrize("dims", [(1, 4, 6, 10), (2, 6, 8, 20), (1, 2, 10, 34)])
def test_conv_stride2_vs_torch(cin, cout, dims, arith):
    b, d, h, w = dims
    x = seeded((b, cin, d, h, w), 3)
    wt = seeded((cout, cin, 3, 3, 3), 4, -0.2, 0.2)
    ref = F.conv3d(x, wt, stride=2, padding=1)
    out = conv3d.conv_plain(cl(x), wt.to(DEV), conv3d.CONV_S2, arith)
    close(ncdhw(out), ref, 1e-4, 2e-5)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 32), (32, 32), (32, 64)])
@pytest.mark.parametrize("dims", [(1, 2, 3, 5), (2, 3, 4, 17), (1, 1, 5, 9)])
def test_deconv_stride2_vs_torch(cin, cout, dims, arith):
    b, d, h, w = dims
    x = seeded((b, cin, d, h, w), 5)
    wt = seeded((cin, cout, 3, 3, 3), 6, -0.2, 0.2)
    ref = F.conv_transpose3d(x, wt, stride=2, padding=1, output_padding=1)
    out = conv3d.conv_plain(cl(x), wt.to(DEV), conv3d.DECONV_S2, arith)
    close(ncdhw(out), ref, 1e-4, 2e-5)


@pytest.mark.parametrize("cin,cout,stride", [(64, 32, 1), (32, 32, 1), (32, 64, 2), (64, 64, 2), (64, 64, 1)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_convbn3d_golden(golden, cin, cout, stride, mode, arith):
    g = golden("g3_convbn3d")
    tag = f"cb{cin}_{cout}_{stride}_{mode}"
    unit = load_procedural(psmnet_3.convbn_3d(cin, cout, 3, stride, 1), f"g3.cb{cin}_{cout}_{stride}.").to(DEV)
    unit.train(mode == "train")
    x = seeded((2, cin, 4, 6, 8), 310 + cin + cout + stride)
    xg = cl(x).requires_grad_(mode == "train")
    if mode == "eval":
        with torch.no_grad():
            y = agg3d.conv_bn(xg, unit, arith=arith)
        close(ncdhw(y), g[tag + "_y"], 1e-4, 2e-5)
        return
    y = agg3d.conv_bn(xg, unit, arith=arith)
    close(ncdhw(y), g[tag + "_y"], 1e-4, 5e-5)
    ct = seeded(tuple(g[tag + "_y"].shape), 320)
    y.backward(cl(ct))
    close(ncdhw(xg.grad), g[tag + "_gx"], 1e-3, 1e-4)
    close(unit[0].weight.grad[:8, :8], g[tag + "_gw"], 1e-3, 2e-4)
    close(unit[1].weight.grad, g[tag + "_gg"], 1e-3, 1e-3)
    close(unit[1].bias.grad, g[tag + "_gb"], 1e-3, 1e-3)


@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True), (False, True)])
def test_convbn_residual_relu_train_vs_oracle(relu, with_res, arith):
    torch.manual_seed(3)
    ref_unit = load_procedural(po._cb3(32, 32, 1), "t.cb.")
    unit = load_procedural(psmnet_3.convbn_3d(32, 32, 3, 1, 1), "t.cb.").to(DEV)
    x, res = seeded((2, 32, 5, 6, 18), 11), seeded((2, 32, 5, 6, 18), 12)
    ct = seeded((2, 32, 5, 6, 18), 13)
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    yr = ref_unit(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(ct)
    xg, rg = cl(x).requires_grad_(), cl(res).requires_grad_()
    y = agg3d.conv_bn(xg, unit, relu=relu, add=rg if with_res else None, arith=arith)
    close(ncdhw(y), yr, 1e-4, 5e-5)
    y.backward(cl(ct))
    close(ncdhw(xg.grad), xr.grad, 1e-3, 1e-4)
    if with_res:
        close(ncdhw(rg.grad), rr.grad, 1e-5, 1e-6)
    close(unit[0].weight.grad, ref_unit[0].weight.grad, 1e-3, 3e-4)
    close(unit[1].weight.grad, ref_unit[1].weight.grad, 1e-3, 1e-3)
    close(unit[1].bias.grad, ref_unit[1].bias.grad, 1e-3, 1e-3)
    close(unit[1].running_mean, ref_unit[1].running_mean, 1e-5, 1e-6)
    close(unit[1].running_var, ref_unit[1].running_var, 1e-5, 1e-6)
    assert int(unit[1].num_batches_tracked) == 1


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 32)])
def test_deconvbn_train_vs_oracle(cin, cout, arith):
    ref_unit = load_procedural(po._up3(cin, cout), "t.up.")
    unit = load_procedural(psmnet_3._up_unit(cin, cout), "t.up.").to(DEV)
    x, res = seeded((2, cin, 3, 4, 10), 21), seeded((2, cout, 6, 8, 20), 22)
    ct = seeded((2, cout, 6, 8, 20), 23)
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    yr = F.relu(ref_unit(xr) + rr)
    yr.backward(ct)
    xg, rg = cl(x).requires_grad_(), cl(res).requires_grad_()
    y = agg3d.deconv_bn(xg, unit, relu=True, add=rg, arith=arith)
    close(ncdhw(y), yr, 1e-4, 5e-5)
    y.backward(cl(ct))
    close(ncdhw(xg.grad), xr.grad, 1e-3, 1e-4)
    close(ncdhw(rg.grad), rr.grad, 1e-5, 1e-6)
    close(unit[0].weight.grad, ref_unit[0].weight.grad, 1e-3, 3e-4)
    close(unit[1].weight.grad, ref_unit[1].weight.grad, 1e-3, 1e-3)


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("skips", [False, True])
def test_hourglass_golden(golden, mode, skips, arith):
    g = golden("g3_hourglass")
    hg = load_procedural(psmnet_3.hourglass(32), "g3.hg.").to(DEV)
    hg.train(mode == "train")
    x = cl(seeded((1, 32, 8, 8, 12), 301)).requires_grad_(mode == "train")
    pre, post = cl(seeded((1, 64, 4, 4, 6), 302)), cl(seeded((1, 64, 4, 4, 6), 303))
    tag = f"{mode}_{'skip' if skips else 'noskip'}"
    if mode == "eval":
        with torch.no_grad():
            o, p, q = hg(x, pre if skips else None, post if skips else None, arith=arith)
    else:
        o, p, q = hg(x, pre if skips else None, post if skips else None, arith=arith)
    close(ncdhw(o), g[tag + "_out"], 1e-3, 1e-4)
    close(ncdhw(p), g[tag + "_pre"], 1e-3, 1e-4)
    close(ncdhw(q), g[tag + "_post"], 1e-3, 1e-4)
    if mode == "train":
        ((o * cl(seeded((1, 32, 8, 8, 12), 304))).sum() + p.sum() * 0.25 + q.sum() * 0.5).backward()
        close(ncdhw(x.grad), g[tag + "_gx"], 2e-3, 2e-4)
        close(hg.conv1[0][0].weight.grad[:8, :8], g[tag + "_gw_conv1"], 2e-3, 5e-4)
        close(hg.conv5[0].weight.grad[:8, :8], g[tag + "_gw_conv5"], 2e-3, 5e-4)
        close(hg.conv6[0].weight.grad[:8, :8], g[tag + "_gw_conv6"], 2e-3, 5e-4)
        close(hg.conv2[1].weight.grad, g[tag + "_ggamma_conv2"], 2e-3, 2e-3)
        close(hg.conv6[1].bias.grad, g[tag + "_gbeta_conv6"], 2e-3, 2e-3)
        close(hg.conv1[0][1].running_mean, g[tag + "_rm_conv1"], 1e-4, 1e-5)
        close(hg.conv1[0][1].running_var, g[tag + "_rv_conv1"], 1e-4, 1e-5)


@pytest.mark.parametrize("dims", [(1, 3, 5, 7), (2, 6, 17, 40), (1, 4, 8, 33)])
def test_classifier_conv_vs_torch(dims):
    b, d, h, w = dims
    x = seeded((b, 32, d, h, w), 31)
    wt = seeded((1, 32, 3, 3, 3), 32, -0.3, 0.3)
    addend = seeded((b, d, h, w), 33)
    ct = seeded((b, d, h, w), 34)
    xr, wr, ar = x.clone().requires_grad_(), wt.clone().requires_grad_(), addend.clone().requires_grad_()
    yr = F.conv3d(xr, wr, padding=1)[:, 0] + ar
    yr.backward(ct)
    conv = torch.nn.Conv3d(32, 1, 3, padding=1, bias=False).to(DEV)
    conv.weight.data.copy_(wt)
    xg, ag = cl(x).requires_grad_(), addend.to(DEV).requires_grad_()
    y = conv3d.conv_logits(xg, conv, ag)
    close(y, yr, 1e-4, 2e-5)
    y.backward(ct.to(DEV))
    close(ncdhw(xg.grad), xr.grad, 1e-4, 2e-5)
    close(conv.weight.grad, wr.grad, 1e-3, 2e-4)
    close(ag.grad, ar.grad, 0, 0)
    y2 = conv3d.conv_logits(xg, conv, None)
    close(y2, F.conv3d(x, wt, padding=1)[:, 0], 1e-4, 2e-5)


def test_add_and_layout_roundtrip():
    a, b = seeded((1, 32, 3, 4, 6), 41), seeded((1, 32, 3, 4, 6), 42)
    y = conv3d.add(cl(a), cl(b))
    close(ncdhw(y), a + b, 0, 0)


def test_fused_cost_volume_conv_equals_materialised(arith):
    """dres0[0] in eval mode: operand synthesised in-kernel (src=1) vs the materialised
    NDHWC volume (K3) through the same MFMA kernel -- must agree bit for bit; and both
    against the oracle's conv on the oracle's volume."""
    b, h, w, nd = 2, 10, 40, 12
    fl, fr = seeded((b, 32, h, w), 51), seeded((b, 32, h, w), 52)
    unit = load_procedural(psmnet_3.convbn_3d(64, 32, 3, 1, 1), "t.d0.").to(DEV).eval()
    ref_unit = load_procedural(po._cb3(64, 32, 1), "t.d0.").eval()
    with torch.no_grad():
        ref = F.relu(ref_unit(po.build_cost_volume(fl, fr, nd)))
        lazy = agg3d.volume_from_features(fl.to(DEV), fr.to(DEV), nd, lazy=True)
        assert isinstance(lazy, conv3d.LazyCostVolume)
        y_fused = agg3d.conv_bn(lazy, unit, relu=True, arith=arith)
        vol = agg3d.volume_from_features(fl.to(DEV), fr.to(DEV), nd, lazy=False)
        y_mat = agg3d.conv_bn(vol, unit, relu=True, arith=arith)
    # (two kernels since round 3: the fused operand runs on the 32x32x16 kernel, the materialised volume on the
    #  depth-rolling 16x16x32 one -- same arithmetic, K blocks of 16 vs 32: equal to rounding, not to the bit)
    assert torch.allclose(y_fused, y_mat, rtol=1e-5, atol=2e-6 * float(y_mat.abs().max()))
    close(ncdhw(y_fused), ref, 1e-4, 2e-5)


def test_convbn_eval_mode_backward_vs_oracle(arith):
    """Frozen-BatchNorm fine-tuning: eval-mode units under autograd (the reference supports it through
    plain nn.Modules); forward and every gradient against the oracle's modules."""
    ref_unit = load_procedural(po._cb3(32, 32, 1), "t.cbe.").eval()
    unit = load_procedural(psmnet_3.convbn_3d(32, 32, 3, 1, 1), "t.cbe.").to(DEV).eval()
    x, res = seeded((2, 32, 4, 6, 18), 81), seeded((2, 32, 4, 6, 18), 82)
    ct = seeded((2, 32, 4, 6, 18), 83)
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    yr = F.relu(ref_unit(xr) + rr)
    yr.backward(ct)
    xg, rg = cl(x).requires_grad_(), cl(res).requires_grad_()
    y = agg3d.conv_bn(xg, unit, relu=True, add=rg, arith=arith)
    close(ncdhw(y), yr, 1e-4, 5e-5)
    y.backward(cl(ct))
    close(ncdhw(xg.grad), xr.grad, 1e-3, 1e-4)
    close(ncdhw(rg.grad), rr.grad, 1e-5, 1e-6)
    close(unit[0].weight.grad, ref_unit[0].weight.grad, 1e-3, 3e-4)
    close(unit[1].weight.grad, ref_unit[1].weight.grad, 1e-3, 1e-3)
    close(unit[1].bias.grad, ref_unit[1].bias.grad, 1e-3, 1e-3)
    assert int(unit[1].num_batches_tracked) == 0


# Stride-2 weight gradients (conv1 / conv3 / conv5 / conv6 of an hourglass) against torch's fp64 gradient.  The f16x3 route of
# the 64-channel-coarse shapes is az_conv3d_wgrad16s2.hip: 4 coarse rows x 8 positions per step, one workgroup per CU walking
# several columns -- shapes with ragged rows / chunks, odd fine sizes, one row, and more columns than workgroups.
@pytest.mark.parametrize("kind,cin,cout", [("conv", 32, 64), ("conv", 64, 64), ("deconv", 64, 32), ("deconv", 64, 64), ("conv", 32, 32)])
@pytest.mark.parametrize("fine_dims", [(1, 4, 8, 16), (2, 6, 10, 36), (1, 3, 7, 13), (1, 2, 2, 50), (2, 24, 12, 192), (1, 5, 34, 20)])
@pytest.mark.parametrize("prec", ["f16x3", "bf16x6", "fp32"])
def test_weight_grad_stride2_vs_torch(kind, cin, cout, fine_dims, prec):
    precision = {"f16x3": conv3d.F16X3, "bf16x6": conv3d.BF16X6, "fp32": conv3d.FP32}[prec]
    if prec != "f16x3" and fine_dims[2] > 10 and fine_dims != (1, 5, 34, 20):
        pytest.skip("the large multi-column shape is there for the f16x3 kernel's column walk")
    b, df, hf, wf = fine_dims
    dc, hc, wc = (df + 1) // 2, (hf + 1) // 2, (wf + 1) // 2
    if kind == "conv":
        x = seeded((b, cin, df, hf, wf), 31).double().requires_grad_(False)
        w = seeded((cout, cin, 3, 3, 3), 32, -0.2, 0.2).double().requires_grad_(True)
        dy = seeded((b, cout, dc, hc, wc), 33)
        F.conv3d(x, w, stride=2, padding=1).backward(dy.double())
        got = conv3d._weight_grad(cl(x.float()), cl(dy), conv3d.CONV_S2, cin, cout, precision)
    else:
        if df % 2 or hf % 2 or wf % 2:
            pytest.skip("output_padding = 1 always gives even fine sizes")
        x = seeded((b, cin, dc, hc, wc), 34).double()
        w = seeded((cin, cout, 3, 3, 3), 35, -0.2, 0.2).double().requires_grad_(True)
        dy = seeded((b, cout, df, hf, wf), 36)
        F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1).backward(dy.double())
        got = conv3d._weight_grad(cl(x.float()), cl(dy), conv3d.DECONV_S2, cin, cout, precision)
    ref = w.grad
    err = (got.detach().cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * ref.abs().max().item() + 1e-6, err


# ---- round 5: the BatchNorm-backward apply pass writes d(raw) pre-split (include/azhip.h "S2 format") --------------------
def _decode_split(t, amax):
    """a pre-split tensor -> fp64 values: every 16 bytes = hi(c0) hi(c1) | hi(c2) hi(c3) | lo(c0) lo(c1) | lo(c2) lo(c3)"""
    a = float(amax[::64].max())  # (the slots of an amax array: every 64th float; the words between are not defined)
    e = int(np.floor(np.log2(a))) if a > 0 else -127
    k = min(max(14 - e, -126), 127)
    h = t.contiguous().view(torch.int16).view(-1, 8).cpu()  # 8 halves per group of four channels
    f = h.view(torch.float16).double()
    vals = f[:, :4] + f[:, 4:]
    return (vals * 2.0 ** (-k)).view(t.shape)


@pytest.mark.parametrize("c,shape", [(32, (2, 5, 9, 20)), (64, (1, 4, 7, 12)), (32, (1, 24, 40, 96))])
@pytest.mark.parametrize("relu", [False, True])
def test_bn_backward_presplit_output_is_the_split_of_the_fp32_one(c, shape, relu, capsys):
    from activezero_amd import _lib
    from activezero_amd.ops import _call, _p, _stream
    b, d, h, w = shape
    raw = cl(seeded((b, c, d, h, w), 61) * 3.0 + 0.5)
    gy = cl(seeded((b, c, d, h, w), 62) * 1e-3)
    gy[0, 1, 2, 3, :] *= 300.0  # a few spikes: the bound must follow the channel that has them
    mean = raw.mean(dim=(0, 1, 2, 3)).contiguous()
    invstd = (raw.var(dim=(0, 1, 2, 3), unbiased=False) + 1e-5).rsqrt().contiguous()
    gamma = (seeded((c,), 63) * 0.5 + 1.0).to(DEV)
    beta = (seeded((c,), 64) * 0.3).to(DEV)
    scale, shift = (gamma * invstd).contiguous(), (beta - mean * gamma * invstd).contiguous()
    nvox = raw.numel() // c
    lib = _lib.lib()
    wsb = lib.az_bn3d_bwd_workspace(nvox, c)
    out = {}
    for split in (0, 1):
        ws = torch.empty(wsb // 4, device=DEV)
        dx = torch.empty_like(raw)
        dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(c, 3, device=DEV)
        am = torch.full((conv3d.AMAX_SLOTS,), 7.0, device=DEV)  # (need not be zero: the first kernel clears it)
        _call("az_bn3d_bwd", _p(dx), None, _p(dg), _p(db), _p(coef), _p(ws), wsb, _p(gy), None, _p(raw), _p(mean), _p(invstd),
              _p(gamma), _p(scale) if relu else None, _p(shift) if relu else None, int(relu), nvox, c, _p(am), split, _stream())
        out[split] = (dx, am, dg, db)
    ref, am0 = out[0][0].double().cpu(), float(out[0][1][::64].max())
    got = _decode_split(out[1][0], out[1][1])
    bound = float(out[1][1][::64].max())
    assert abs(am0 - float(ref.abs().max())) == 0.0
    assert bound >= am0, (bound, am0)
    with capsys.disabled():
        print(f"\npre-split dx C={c} {shape} relu={relu}: bound / max|dx| = {bound / am0:.3f}")
    assert bound <= 4.0 * am0  # (a loose bound costs dynamic range, include/azhip.h: log2 of this ratio in bits)
    # hi + lo = x up to 2^-22 |x|, and the fp16 subnormal spacing 2^-25 on the scaled value = 2^-39 of the bound (x2: margin)
    err = (got - ref).abs()
    assert float((err - (2.0 ** -21 * ref.abs() + 2.0 ** -38 * bound)).max()) <= 0.0
    # the parameter gradients do not depend on the output format
    torch.testing.assert_close(out[1][2], out[0][2], rtol=0, atol=0)
    torch.testing.assert_close(out[1][3], out[0][3], rtol=0, atol=0)
    # ... and the consumers multiply exactly these bits: input gradient from the pre-split tensor == from its decoded floats
    if c == 32:
        wt = (seeded((c, c, 3, 3, 3), 65) * 0.1).to(DEV)
        dxs = out[1][0]
        conv3d._set_amax(dxs, out[1][1])
        dxs.az_split = True
        dec = got.float().to(DEV)
        conv3d._set_amax(dec, out[1][1])  # the same scale: the same two fp16 parts
        with torch.no_grad():
            a = conv3d._input_grad(dxs, wt, conv3d.CONV_S1, c, c, conv3d.F16X3)
            bb = conv3d._input_grad(dec, wt, conv3d.CONV_S1, c, c, conv3d.F16X3)
            # (not bit-equal: hi + lo of an element may span more than fp32's 24 bits, so the decoded float re-splits to a
            #  lo one ulp off -- 2^-24-sized differences of single products)
            torch.testing.assert_close(a, bb, rtol=0, atol=2.0 ** -20 * float(bb.abs().max()))
            xin = cl(seeded((b, c, d, h, w), 66))
            ga = conv3d._weight_grad(xin, dxs, conv3d.CONV_S1, c, c, conv3d.F16X3)
            gb = conv3d._weight_grad(xin, dec, conv3d.CONV_S1, c, c, conv3d.F16X3)
            torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-6 * float(gb.abs().max()))  # (float-atomic flush order)


def test_presplit_routing_matches_the_library_queries():
    """which layers get a pre-split d(raw): both readers must stage it by copy (conv3d._presplit_ok)"""
    def t(c, d, h, w):
        return torch.empty(1, d, h, w, c, device=DEV)
    ok = conv3d._presplit_ok
    assert ok(t(32, 8, 16, 32), t(32, 8, 16, 32), conv3d.CONV_S1, 32, 32, True, True)      # V0 layers: roll + wgrad_r16
    assert ok(t(32, 8, 16, 32), t(64, 4, 8, 16), conv3d.CONV_S2, 32, 64, True, True)       # conv1: t2roll + wgrad_s2r16
    assert ok(t(64, 4, 8, 16), t(64, 4, 8, 16), conv3d.CONV_S1, 64, 64, False, True)       # weight gradient alone
    assert ok(t(64, 4, 8, 16), t(32, 8, 16, 32), conv3d.DECONV_S2, 64, 32, False, True)    # transposed: fine operand
    # gather-kernel input gradients read fp32; the stride-1 64 -> 64 ones moved onto the rolling kernel in round 5 (COUT = 64)
    on_roll64 = conv3d._lib.lib().az_option(b"AZ_CONV_ROLL64") != 0
    assert ok(t(64, 4, 8, 16), t(64, 4, 8, 16), conv3d.CONV_S1, 64, 64, True, True) == on_roll64
    assert not ok(t(64, 4, 8, 16), t(64, 2, 4, 8), conv3d.CONV_S2, 64, 64, True, True)        # conv3: its input gradient is the gather kernel's
    # conv6 (transposed 64 -> 32): its input gradient is the stride-2 rolling kernel since az_conv3d_s2roll.hip
    on_s2roll = conv3d._lib.lib().az_option(b"AZ_CONV_S2ROLL") != 0
    assert ok(t(64, 4, 8, 16), t(32, 8, 16, 32), conv3d.DECONV_S2, 64, 32, True, True) == on_s2roll
    assert not ok(t(64, 2, 4, 8), t(64, 4, 8, 16), conv3d.DECONV_S2, 64, 64, True, True)     # conv5: gather kernel


# ---- round 5: gradient hand-over between the consumers of one tensor (conv3d.GradSlot) -----------------------------------
def _fork_net(x, units, arith):
    """y = unit_a(t) + relu-free residual(t), t = unit0(x): t has two consumers inside one _ConvBN (x and residual of different
    nodes) and a third one through a second branch -- the shapes of the hourglass skips"""
    t = agg3d.conv_bn(x, units[0], relu=True, arith=arith)
    u = agg3d.conv_bn(t, units[1], relu=True, arith=arith)           # consumer 1 of t (input)
    v = agg3d.conv_bn(u, units[2], add=t, arith=arith)               # consumer 2 of t (residual)
    w = agg3d.conv_bn(t, units[3], relu=True, arith=arith)           # consumer 3 of t (input)
    return v, w, t


@pytest.mark.parametrize("use", ["all", "first_only", "second_only"])
def test_gradient_handover_equals_the_engines_sum(use, monkeypatch):
    arith = conv3d.Arith.of("f16x3")
    units = [load_procedural(psmnet_3.convbn_3d(32, 32, 3, 1, 1), f"t.ho{i}.").to(DEV).train() for i in range(4)]
    x0 = seeded((1, 32, 4, 8, 16), 21)
    cv, cw = cl(seeded((1, 32, 4, 8, 16), 22)), cl(seeded((1, 32, 4, 8, 16), 23))

    def run(handover):
        monkeypatch.setattr(conv3d, "_HANDOVER", handover)
        for u in units:
            u.zero_grad(set_to_none=True)
        x = cl(x0).requires_grad_()
        v, w, t = _fork_net(x, units, arith)
        if handover:
            slot = getattr(t, "az_gslot", None)
            assert slot is not None and slot.expect == 3
        # "first_only" / "second_only": one branch never reaches backward -- consumers registered in forward that the engine
        # prunes; the producing node must add what the others parked (the safety net)
        loss = {"all": (v * cv).sum() + (w * cw).sum(), "first_only": (v * cv).sum(), "second_only": (w * cw).sum()}[use]
        loss.backward()
        return [x.grad.clone()] + [p.grad.clone() for u in units for p in u.parameters() if p.grad is not None]

    got, ref = run(True), run(False)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        # same kernels, one addition moved from the engine into an epilogue: float-associativity-sized differences
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-9)
