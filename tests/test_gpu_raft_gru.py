"""the RAFT-Stereo ConvGRU update (reference nets/raft/update.py:19-41) on the
plain-bf16 MFMA convolution with the gate arithmetic in its epilogue.

Pinned by G12 (tests/golden/g12_convgru.npz, tools/make_goldens.py g12_convgru): the reference class itself,
imported from nets/raft/update.py with inert stand-ins for the two absent packages its module imports but the
class never touches, evaluated in fp32, fp64 and under CPU autocast(bfloat16) -- test_gru_matches_reference_golden
and test_gru_autograd_matches_reference_golden below.  The 22 lines are also restated in torch here for the
shape sweeps.  Checked besides:
  * the convolution kernel itself, bit-level: with operands that are exactly representable in bf16 and
    products that sum exactly in fp32, the result equals an fp64 convolution;
  * the GRU against an fp64 evaluation of the reference's formula, next to the error of the reference's own
    arithmetic (the same formula under CPU autocast(bfloat16)): ours must not be worse;
  * the autograd path (bf16x6 kernels) against torch fp32, values and gradients.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference_gru(mod, h, cz, cr, cq, *x_list):
    # nets/raft/update.py:32-41
    x = torch.cat(x_list, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(mod.convz(hx) + cz)
    r = torch.sigmoid(mod.convr(hx) + cr)
    q = torch.tanh(mod.convq(torch.cat([r * h, x], dim=1)) + cq)
    return (1 - z) * h + z * q


class _TorchGRU(nn.Module):
    def __init__(self, hidden_dim, input_dim):
        super().__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)


def _inputs(b, c, cx, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    hid = torch.tanh(torch.randn(b, c, h, w, generator=g))
    ctx = [torch.randn(b, c, h, w, generator=g) * 0.5 for _ in range(3)]
    xs = [torch.randn(b, n, h, w, generator=g) for n in cx]
    return hid, ctx, xs


def test_bf16_conv_exact_on_representable_operands():
    from activezero_amd.nets.raft import gru
    g = torch.Generator().manual_seed(3)
    b, h, w, cin, cout = 2, 19, 37, 48, 96
    x = torch.randint(-8, 9, (b, cin, h, w), generator=g).float()          # small integers: exact in bf16,
    wt = torch.randint(-4, 5, (cout, cin, 3, 3), generator=g).float() / 8  # products and sums exact in fp32
    bias = torch.randn(cout, generator=g)
    res = torch.randn(b, cout, h, w, generator=g)
    want = F.conv2d(x.double(), wt.double(), padding=1) + bias.double().view(1, -1, 1, 1) + res.double()
    xr = x.permute(0, 2, 3, 1).contiguous().cuda()
    pk = gru._pack_bf16((wt.cuda(),), False)
    for act, fn in ((gru.ACT_NONE, lambda t: t), (gru.ACT_RELU, torch.relu), (gru.ACT_SIGMOID, torch.sigmoid),
                    (gru.ACT_TANH, torch.tanh)):
        got = gru.conv3x3_bf16(xr, pk, cin, cout, bias.cuda(), res.permute(0, 2, 3, 1).contiguous().cuda(), act)
        ref = fn(want).permute(0, 2, 3, 1)
        tol = 0 if act in (gru.ACT_NONE, gru.ACT_RELU) else 2e-6
        err = (got.cpu().double() - ref).abs().max().item()
        lim = tol if tol else 4e-6 * want.abs().max().item()  # bias / residual adds round once each in fp32
        assert err <= lim, (act, err)


@pytest.mark.parametrize("hidden,cx,hw", [(128, (128,), (34, 60)), (128, (128, 128), (17, 30)), (64, (32, 48), (9, 21))])
def test_gru_matches_reference_formula(hidden, cx, hw):
    from activezero_amd.nets.raft.gru import ConvGRU
    torch.manual_seed(11)
    ref = _TorchGRU(hidden, sum(cx))
    mod = ConvGRU(hidden, sum(cx))
    mod.load_state_dict(ref.state_dict())  # same parameter names as the reference class
    hid, ctx, xs = _inputs(2, hidden, cx, *hw, seed=5)
    with torch.no_grad():
        exact = _reference_gru(ref.double(), hid.double(), *[c.double() for c in ctx], *[x.double() for x in xs])
        ref.float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            amp = _reference_gru(ref, hid, *ctx, *xs).float()
        mod.cuda()
        got = mod(hid.cuda(), *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
        with torch.autocast("cuda", dtype=torch.bfloat16):  # what raft_stereo.py:142 wraps the call in
            got_amp = mod(hid.cuda(), *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
    assert got.shape == exact.shape and got.dtype == torch.float32
    e_hip = (got.cpu().double() - exact).abs().max().item()
    e_amp = (amp.double() - exact).abs().max().item()
    assert torch.equal(got, got_amp)
    assert e_hip <= 2e-2, e_hip            # bf16 operand rounding over K = 9 * (hidden + input) terms
    assert e_hip <= e_amp, (e_hip, e_amp)  # no worse than the reference's autocast arithmetic
    # iterating feeds the state back (raft_stereo.py:138-172): stays bounded and close after 4 updates
    with torch.no_grad():
        hh, he = hid.cuda(), hid.double()
        ref.double()
        for _ in range(4):
            hh = mod(hh, *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
            he = _reference_gru(ref, he, *[c.double() for c in ctx], *[x.double() for x in xs])
    assert (hh.cpu().double() - he).abs().max().item() <= 5e-2


def test_gru_autograd_path():
    from activezero_amd.nets.raft.gru import ConvGRU
    torch.manual_seed(2)
    ref = _TorchGRU(64, 64)
    mod = ConvGRU(64, 64)
    mod.train_arithmetic = "bf16x6"  # (the fp32-class route; the default, the reference's autocast arithmetic, is pinned by G12)
    mod.load_state_dict(ref.state_dict())
    mod.cuda()
    hid, ctx, xs = _inputs(1, 64, (64,), 12, 20, seed=9)
    cot = torch.randn(1, 64, 12, 20)
    hr = hid.clone().requires_grad_(True)
    (_reference_gru(ref, hr, *ctx, *xs) * cot).sum().backward()
    hg = hid.cuda().requires_grad_(True)
    out = mod(hg, *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
    (out * cot.cuda()).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), _reference_gru(ref, hid, *ctx, *xs).detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(hg.grad.cpu(), hr.grad, rtol=1e-4, atol=1e-5)
    for name in ("convz", "convr", "convq"):
        torch.testing.assert_close(getattr(mod, name).weight.grad.cpu(), getattr(ref, name).weight.grad,
                                   rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(getattr(mod, name).bias.grad.cpu(), getattr(ref, name).bias.grad,
                                   rtol=1e-4, atol=2e-5)


def _g12_case(g, tag):
    from tests._weights import load_procedural, seeded
    from activezero_amd.nets.raft.gru import ConvGRU
    meta = [int(v) for v in g[f"{tag}_meta"]]
    hidden, h, w, b, sd, cs, ps = meta[:7]
    cx = meta[7:]
    mod = load_procedural(ConvGRU(hidden, sum(cx)), f"g12{tag}.").cuda()  # the reference's parameter names
    hid = torch.tanh(seeded((b, hidden, h, w), sd, -2.0, 2.0))
    ctx = [seeded((b, hidden, h, w), sd + 1 + i, -0.8, 0.8) for i in range(3)]
    xs = [seeded((b, n, h, w), sd + 4 + i, -1.7, 1.7) for i, n in enumerate(cx)]
    lat = lambda t: t[:, ::cs, ::ps, ::ps].detach().cpu().double().numpy()
    return mod, hid, ctx, xs, lat, (b, hidden, h, w, sd)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gru_matches_reference_golden(golden, tag, capsys):
    """ConvGRU.forward of the imported reference (update.py:19-41).  e64 = max |. - ref64|: the no-grad bf16
    path must be no further from the exact result than the reference's own autocast(bfloat16) arithmetic is,
    and within the spread that arithmetic has around its fp32 evaluation; one update and four chained ones."""
    g = golden("g12_convgru")
    mod, hid, ctx, xs, lat, _ = _g12_case(g, tag)
    args = [c.cuda() for c in ctx] + [x.cuda() for x in xs]
    with torch.no_grad():
        got = mod(hid.cuda(), *args)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got_amp = mod(hid.cuda(), *args)
        hh = hid.cuda()
        for _ in range(4):
            hh = mod(hh, *args)
    assert torch.equal(got, got_amp)
    r32, r64, ramp = g[f"{tag}_out32"].astype("f8"), g[f"{tag}_out64"], g[f"{tag}_outamp"].astype("f8")
    e_hip64, e_amp64 = abs(lat(got) - r64).max(), abs(ramp - r64).max()
    e_hip32, e_amp32 = abs(lat(got) - r32).max(), abs(ramp - r32).max()
    m_hip64, m_amp64 = abs(lat(got) - r64).mean(), abs(ramp - r64).mean()
    e_it, e_it_ref = abs(lat(hh) - g[f"{tag}_iter4_64"]).max(), abs(g[f"{tag}_iter4_32"].astype("f8") - g[f"{tag}_iter4_64"]).max()
    with capsys.disabled():
        print(f"\nG12 {tag}: max|hip-ref64| {e_hip64:.2e} (reference autocast: {e_amp64:.2e}; means {m_hip64:.2e} / {m_amp64:.2e})  max|hip-ref32| {e_hip32:.2e} "
              f"(autocast-ref32 {e_amp32:.2e})  4 updates: {e_it:.2e} (ref32-ref64 {e_it_ref:.1e})")
    # mean error no larger than that of the reference's autocast arithmetic; the maximum over ~1e5 values is a noisy
    # statistic (both round every operand to 8 bits): 25 % margin on it; and a fixed ceiling (measured 1.2e-2 / 1.6e-2)
    assert m_hip64 <= m_amp64, (m_hip64, m_amp64)
    assert e_hip64 <= 1.25 * e_amp64 and e_hip32 <= 1.25 * e_amp32, (e_hip64, e_amp64, e_hip32, e_amp32)
    assert e_hip64 <= 2e-2 and e_it <= 4e-2, (e_hip64, e_it)


@pytest.mark.parametrize("mode,suffix", [("f16x1", "amp16"), ("bf16", "amp")])
def test_gru_training_arithmetic_matches_reference_autocast(golden, capsys, mode, suffix):
    """The TRAINING route in the reference's arithmetic (update block under autocast, raft_stereo.py:142-172,
    train.py:303-309): forward, input gradient and weight gradient with operands rounded ONCE to 16 bits.  The reference's
    torch.cuda.amp.autocast is FLOAT16 on CUDA: G12 holds the gradients of the imported class under CPU autocast(float16)
    (`*_amp16`, round 5) next to the bfloat16 ones of round 4 (`*_amp`) and the fp64 ones; e_amp = |g_amp - g64| / |g64| is
    how far the reference's own 16-bit evaluation sits from the exact gradient (fp16: 3-7e-4; bf16: 2.5-5.6e-3).
      * "f16x1" (default): one fp16 part per operand -- must be at least as close to the exact gradient as the reference's
        FLOAT16 evaluation (this path keeps the gates and the state in fp32);
      * "bf16" (round 4's default): held to the bfloat16 yardstick only -- 8x coarser operand rounding than the reference's,
        which is why it is no longer the default."""
    from tests._weights import seeded
    g = golden("g12_convgru")
    mod, hid, ctx, xs, lat, (b, hidden, h, w, sd) = _g12_case(g, "d")
    assert mod.train_arithmetic == "f16x1"
    mod.train_arithmetic = mode
    hg = hid.cuda().requires_grad_(True)
    out = mod(hg, *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
    (out * seeded((b, hidden, h, w), sd + 9).cuda()).sum().backward()
    # value: no further from fp64 than the reference's autocast forward
    key_out = "d_outamp16" if suffix == "amp16" else "d_outamp"
    e_out, e_out_amp = abs(lat(out) - g["d_out64"]).max(), abs(g[key_out].astype("f8") - g["d_out64"]).max()
    assert e_out <= 1.1 * e_out_amp, (e_out, e_out_amp)
    rel = lambda a, ref: float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    rows = [("gh", hg.grad.cpu().double().numpy(), g["d_gh64"], g[f"d_gh_{suffix}"].astype("f8"))]
    for name in ("convz", "convr", "convq"):
        rows.append((name + ".weight", getattr(mod, name).weight.grad.cpu().double().numpy(), g[f"d_gwfull64_{name}"],
                     g[f"d_gwfull_{suffix}_{name}"].astype("f8")))
        rows.append((name + ".bias", getattr(mod, name).bias.grad.cpu().double().numpy(), g[f"d_gb64_{name}"],
                     g[f"d_gb_{suffix}_{name}"].astype("f8")))
    lines = []
    for name, got, r64, ramp in rows:
        e_hip, e_amp, e_cross = rel(got, r64), rel(ramp, r64), rel(got, ramp)
        lines.append(f"G12 train[{mode}] {name:14s} hip-ref64 {e_hip:.2e}  autocast({'fp16' if suffix == 'amp16' else 'bf16'})-ref64 {e_amp:.2e}  hip-autocast {e_cross:.2e}")
        assert e_hip <= 1.1 * e_amp + 1e-6, lines[-1]
        assert e_cross <= e_hip + e_amp + 1e-6, lines[-1]
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def test_gru_context_rows_converted_once_per_step_with_fp16_inputs():
    """The reference's default config (MIXED_PRECISION, raft_stereo.py:142-172) hands the context terms over as fp16 tensors
    and the same objects in all 22 updates of a step: one conversion per step and level, keyed on the caller's tensors
    (a key taken after `.float()` never hit), and nothing of the step stays alive in the cache afterwards."""
    import gc
    from activezero_amd.nets.raft import gru
    from activezero_amd.nets.raft.gru import ConvGRU
    torch.manual_seed(7)
    b, c, ci, h, w = 1, 32, 96, 14, 22
    mod = ConvGRU(c, ci).cuda()
    hid, ctx, xs = _inputs(b, c, (36, 60), h, w, 91)
    ctx16 = [t.cuda().half().requires_grad_(True) for t in ctx]
    state = hid.cuda().requires_grad_(True)
    n0 = gru.CTX_CONVERSIONS
    with torch.autocast("cuda", dtype=torch.float16):
        s = state
        for _ in range(3):
            s = mod(s, *ctx16, *[x.cuda() for x in xs])
    assert gru.CTX_CONVERSIONS - n0 == 1
    s.float().sum().backward()
    assert all(t.grad is not None and t.grad.dtype == torch.float16 for t in ctx16)
    del ctx16, s
    gc.collect()
    hid2, ctx2, _ = _inputs(b, c, (36, 60), h, w, 92)
    mod(hid2.cuda().requires_grad_(True), *[t.cuda().half() for t in ctx2], *[x.cuda() for x in xs])  # a new step: one more, dead entries dropped
    assert gru.CTX_CONVERSIONS - n0 == 2
    assert len(gru._CTX_CACHE) == 1


def test_gru_autograd_matches_reference_golden(golden):
    """the fp32-class differentiable route (train_arithmetic = "bf16x6") against the reference's fp32 values and gradients"""
    from tests._weights import seeded
    g = golden("g12_convgru")
    mod, hid, ctx, xs, lat, (b, hidden, h, w, sd) = _g12_case(g, "d")
    mod.train_arithmetic = "bf16x6"
    hg = hid.cuda().requires_grad_(True)
    out = mod(hg, *[c.cuda() for c in ctx], *[x.cuda() for x in xs])
    (out * seeded((b, hidden, h, w), sd + 9).cuda()).sum().backward()
    T = torch.from_numpy
    torch.testing.assert_close(out.detach().cpu(), T(g["d_out32"]), rtol=1e-4, atol=1e-5)
    assert abs(lat(out) - g["d_out64"]).max() <= 1.5 * abs(g["d_out32"].astype("f8") - g["d_out64"]).max() + 1e-7
    torch.testing.assert_close(hg.grad.cpu(), T(g["d_gh"]), rtol=1e-4, atol=2e-5)
    for name in ("convz", "convr", "convq"):
        torch.testing.assert_close(getattr(mod, name).weight.grad[:16, :32].cpu(), T(g[f"d_gw_{name}"]), rtol=1e-4, atol=5e-5)
        torch.testing.assert_close(getattr(mod, name).bias.grad.cpu(), T(g[f"d_gb_{name}"]), rtol=1e-4, atol=5e-5)


def test_gru_rejects_cpu_and_bad_channels():
    from activezero_amd.nets.raft.gru import ConvGRU
    mod = ConvGRU(64, 64)
    t = torch.zeros(1, 64, 8, 8)
    with pytest.raises(RuntimeError):
        mod(t, t, t, t, t)
    mod.cuda()
    with pytest.raises(RuntimeError), torch.no_grad():
        mod(t.cuda(), t.cuda(), t.cuda(), t.cuda(), torch.zeros(1, 32, 8, 8).cuda())
    with pytest.raises(RuntimeError):
        ConvGRU(64, 64, kernel_size=5)


def test_gru_fused_training_node_equals_the_operator_form():
    """train_arithmetic "bf16" (one autograd node per update: az_gru_gates.hip between the bf16 convolutions) against "bf16_ops"
    (the same convolutions and torch operators for the gates): same arithmetic, so values and EVERY gradient -- state,
    context terms, inputs, weights, biases -- agree to 2e-4 of each tensor's largest magnitude; two chained
    updates, so the state gradient passes through both routes."""
    from activezero_amd.nets.raft.gru import ConvGRU
    torch.manual_seed(5)
    b, c, ci, h, w = 2, 32, 96, 21, 35
    hid, ctx, xs = _inputs(b, c, (36, 60), h, w, 77)
    res = {}
    for mode in ("bf16", "bf16_ops"):
        torch.manual_seed(11)
        mod = ConvGRU(c, ci).cuda()
        mod.train_arithmetic = mode
        leaves = [hid.clone().cuda().requires_grad_(True)] + [t.clone().cuda().requires_grad_(True) for t in ctx] + \
                 [t.clone().cuda().requires_grad_(True) for t in xs]
        s1 = mod(leaves[0], *leaves[1:4], *leaves[4:])
        s2 = mod(s1, *leaves[1:4], *leaves[4:])
        cot = torch.randn(s2.shape, generator=torch.Generator().manual_seed(3)).cuda()
        ((s2 * cot).sum() + 0.3 * s1.square().sum()).backward()
        res[mode] = [s2.detach()] + [t.grad for t in leaves] + [p.grad for p in mod.parameters()]
    for i, (a, bb) in enumerate(zip(res["bf16"], res["bf16_ops"])):
        err = (a - bb).abs().max().item()
        # (not bit-equal: a gate expression rounded differently in fp32 can move an operand of the NEXT convolution by one bf16
        #  ulp, 2^-9 relative on that element; the arithmetic's own distance from fp64 is 3-6e-3, test above)
        assert err <= 2e-4 * max(bb.abs().max().item(), 1e-3) + 1e-6, (i, err, bb.abs().max().item())


# ---- round 5: the GRU's input rows in one launch (az_rows_concat) and image-layout gradients (az_rows_slice_to_image) ---------
@pytest.mark.parametrize("b,h,w", [(1, 5, 7), (2, 21, 35), (1, 8, 64), (3, 3, 100)])
def test_rows_concat_and_slice_to_image_vs_torch(b, h, w):
    import ctypes
    from activezero_amd.ops import _call, _p, _stream
    g = torch.Generator().manual_seed(b * 100 + h)
    rows_a = torch.randn(b, h, w, 32, generator=g).cuda()          # dense rows
    img_b = torch.randn(b, 36, h, w, generator=g).cuda()           # NCHW image
    rows_c = torch.randn(b, h, w, 220, generator=g).cuda()         # dense rows, more channels than an image source may have
    img_d = torch.randn(b, 128, h, w, generator=g).cuda()          # the largest image source
    want = torch.cat([rows_a, img_b.permute(0, 2, 3, 1), rows_c, img_d.permute(0, 2, 3, 1)], -1)
    got = torch.full_like(want, float("nan"))
    n = 4
    ptrs = (ctypes.c_void_p * n)(rows_a.data_ptr(), img_b.data_ptr(), rows_c.data_ptr(), img_d.data_ptr())
    chans = (ctypes.c_int * n)(32, 36, 220, 128)
    kinds = (ctypes.c_int * n)(0, 1, 0, 1)
    _call("az_rows_concat", _p(got), b * h * w, h * w, n, ptrs, chans, kinds, _stream())
    assert torch.equal(got, want)
    back = torch.full((b, 36, h, w), float("nan"), device="cuda")
    _call("az_rows_slice_to_image", _p(back), _p(got), b * h * w, h * w, 416, 32, 36, _stream())
    assert torch.equal(back, img_b)
    back = torch.full((b, 128, h, w), float("nan"), device="cuda")
    _call("az_rows_slice_to_image", _p(back), _p(got), b * h * w, h * w, 416, 288, 128, _stream())
    assert torch.equal(back, img_d)


def test_gru_update_with_assembled_rows_equals_the_slice_assignments(monkeypatch):
    """the fused training node with its input rows from az_rows_concat (state in channels-last memory from the update before,
    lookup as an NCHW image, context through the cached channels-last copy) against torch's slice assignments: the same bits
    forward and in every gradient; the context copy is made once for the two chained updates"""
    from activezero_amd.nets.raft import gru as G
    torch.manual_seed(5)
    b, c, ci, h, w = 2, 32, 256, 21, 35
    hid, ctx, xs = _inputs(b, c, (36, 220), h, w, 78)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(G, "ASSEMBLE", on)
        torch.manual_seed(11)
        mod = G.ConvGRU(c, ci).cuda()
        leaves = [hid.clone().cuda().requires_grad_(True)] + [t.clone().cuda().requires_grad_(True) for t in ctx] + \
                 [t.clone().cuda().requires_grad_(True) for t in xs]
        before = G.ROW_CONVERSIONS
        s1 = mod(leaves[0], *leaves[1:4], *leaves[4:])
        s2 = mod(s1, *leaves[1:4], *leaves[4:])
        made = G.ROW_CONVERSIONS - before
        # the 220-channel context once (too wide for the kernel's image tile); the 36-channel image goes through the kernel as an
        # image in the first update and -- the SAME tensor again in the second -- gets its cached copy then
        assert made == (2 if on else 0), made
        cot = torch.randn(s2.shape, generator=torch.Generator().manual_seed(3)).cuda()
        ((s2 * cot).sum() + 0.3 * s1.square().sum()).backward()
        res[on] = [s2.detach()] + [t.grad for t in leaves] + [p.grad for p in mod.parameters()]
    for i, (a, bb) in enumerate(zip(res[True], res[False])):
        # (weight gradients go through float atomics: equal to their rounding, everything else to the bit)
        if i < 1 + 6:
            assert torch.equal(a.contiguous(), bb.contiguous()), i
        else:
            torch.testing.assert_close(a, bb, rtol=1e-5, atol=1e-6 * float(bb.abs().max()))
