"""GPU: the pack plan (activezero_amd/conv3d.py PackPlan; include/azhip.h az_pack_f16_multi) -- every f16x3 weight image of a
model written by ONE launch per optimizer step into persistent buffers:
  * the multi-tensor kernel writes, for each of the four layouts (flipped / padded cases included), the bytes the
    per-tensor entry points write;
  * a training run with the plan equals the same run with a launch per image (AZ_PACK_PLAN=0 route), step by step, and
    costs one az_pack_f16_multi launch per step from the second step on;
  * the plan never answers for a dead parameter's address."""
import copy
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from activezero_amd import _lib, conv2d, conv3d  # noqa: E402
from activezero_amd.nets.psmnet import psmnet_3 as psm3  # noqa: E402
from activezero_amd.ops import _call, _p, _stream  # noqa: E402
from oracle import psmnet_oracle as po  # noqa: E402
from tests._weights import load_procedural, seeded  # noqa: E402

DEV = torch.device("cuda", 0)


def test_multi_pack_writes_what_the_per_tensor_launches_write():
    plan = conv3d.PackPlan(DEV)
    cases = []  # (parameter, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip, reference packer)

    def add(w, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip, ref):
        cases.append((torch.nn.Parameter(w.to(DEV)), kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip, ref))

    w3 = seeded((64, 64, 3, 3, 3), 1, -0.3, 0.3)   # Conv3d 64 -> 64, stride 2
    for flip in (False, True):
        # forward image of the stride-2 64 -> 64 layer (gather layout) and the input-gradient image of a stride-1 32 -> 32 layer (roll)
        add(w3, conv3d.PACK_3D_GATHER, 64, 64, 64, 64, 64 * 27, 27, 27, flip,
            lambda pk, w, am, flip=flip: _call("az_conv3d_pack_weights_f16", _p(pk), _p(w), _p(am), 64, 64, 64 * 27, 27, int(flip), conv3d.CONV_S2, _stream()))
    w3s = seeded((64, 32, 3, 3, 3), 6, -0.3, 0.3)  # Conv3d 32 -> 64, stride 2: the rolling layout of az_conv3d_s2roll.hip
    s2_kind = _lib.lib().az_conv3d_f16_layout(conv3d.CONV_S2, 32, 64)
    add(w3s, s2_kind, 32, 64, 32, 64, 32 * 27, 27, 27, False,
        lambda pk, w, am: _call("az_conv3d_pack_weights_f16", _p(pk), _p(w), _p(am), 32, 64, 32 * 27, 27, 0, conv3d.CONV_S2, _stream()))
    r64_kind = _lib.lib().az_conv3d_f16_layout(conv3d.CONV_S1, 64, 64)  # stride-1 64 -> 64: two 32-channel images (AZ_PACK_3D_ROLL2)
    add(w3, r64_kind, 64, 64, 64, 64, 27, 64 * 27, 27, True,
        lambda pk, w, am: _call("az_conv3d_pack_weights_f16", _p(pk), _p(w), _p(am), 64, 64, 27, 64 * 27, 1, conv3d.CONV_S1, _stream()))
    w3b = seeded((32, 32, 3, 3, 3), 2, -0.3, 0.3)
    add(w3b, conv3d.PACK_3D_ROLL, 32, 32, 32, 32, 27, 32 * 27, 27, True,
        lambda pk, w, am: _call("az_conv3d_pack_weights_f16", _p(pk), _p(w), _p(am), 32, 32, 27, 32 * 27, 1, conv3d.CONV_S1, _stream()))
    w2 = seeded((64, 64, 3, 3), 3, -0.3, 0.3)
    add(w2, conv3d.PACK_2D_ROLL, 64, 64, 64, 64, 64 * 9, 9, 9, False,
        lambda pk, w, am: _call("az_conv2d_roll_pack_f16", _p(pk), _p(w), _p(am), 64, 64, 64 * 9, 9, 0, _stream()))
    add(w2, conv3d.PACK_2D_ROLL, 64, 64, 64, 64, 9, 64 * 9, 9, True,
        lambda pk, w, am: _call("az_conv2d_roll_pack_f16", _p(pk), _p(w), _p(am), 64, 64, 9, 64 * 9, 1, _stream()))
    w2p = seeded((128, 320, 3, 3), 4, -0.3, 0.3)   # lastconv's first layer
    add(w2p, conv3d.PACK_2D_SAME, 320, 128, 320, 128, 320 * 9, 9, 9, False,
        lambda pk, w, am: _call("az_conv2d_pack_weights_f16", _p(pk), _p(w), _p(am), 320, 128, 320, 128, 320 * 9, 9, 3, 3, 0, _stream()))
    w1 = seeded((12, 40, 1, 1), 5, -0.3, 0.3)      # channel counts padded to 48 / 32 by the packer
    add(w1, conv3d.PACK_2D_SAME, 48, 32, 40, 12, 40, 1, 1, False,
        lambda pk, w, am: _call("az_conv2d_pack_weights_f16", _p(pk), _p(w), _p(am), 48, 32, 40, 12, 40, 1, 1, 1, 0, _stream()))

    params = [c[0] for c in cases]
    plan.prepack(params)  # registers them (no entries yet: nothing to pack)
    refs = []
    for prm, kind, cin, cout, cir, cor, sco, sci, taps, flip, ref in cases:
        e, fresh = plan.lookup(prm, kind, cin, cout, cir, cor, sco, sci, taps, flip)
        assert e is not None and not fresh
        am = conv3d.absmax(prm.detach())
        want = torch.zeros_like(e.packed)
        ref(want, prm.detach(), am)
        refs.append(want)
        e.packed.fill_(float("nan"))
    assert plan.launches == 0
    plan.prepack(params)  # every entry is stale (version -1): one launch
    assert plan.launches == 1
    for (prm, kind, *_), want, e in zip(cases, refs, plan.table[5]):
        assert torch.equal(e.packed.view(torch.int32), want.view(torch.int32)), kind
        assert e.version == prm._version
    plan.prepack(params)  # nothing moved: no launch
    assert plan.launches == 1
    with torch.no_grad():
        params[0].mul_(3.0)  # an in-place update bumps the version counter: one more launch, new bytes for that weight only
    plan.prepack(params)
    assert plan.launches == 2
    am = conv3d.absmax(params[0].detach())
    want = torch.zeros_like(refs[0])
    cases[0][-1](want, params[0].detach(), am)
    assert torch.equal(plan.table[5][0].packed.view(torch.int32), want.view(torch.int32))
    assert torch.equal(plan.table[5][2].packed.view(torch.int32), refs[2].view(torch.int32))


def _run_steps(model, il, ir, gt, md, n):
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    out = []
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        loss = po.psmnet_disp_loss(model(il, ir), gt, po.disparity_mask(gt, md))
        loss.backward()
        out.append((loss.item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
        opt.step()
    return out


def test_training_with_the_plan_equals_a_launch_per_image(monkeypatch):
    md = 32
    base = load_procedural(psm3.PSMNet(md), "g4.").to(DEV).train().set_weight_grad_overlap(False)
    il, ir = (seeded((1, 3, 256, 256), 820 + i, -2.0, 2.0).to(DEV) for i in range(2))
    gt = 1.0 + 28.0 * seeded((1, 1, 256, 256), 97, 0.0, 1.0).to(DEV)
    plan = conv3d.pack_plan(DEV)
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    l0 = plan.launches
    ra = _run_steps(a, il, ir, gt, md, 3)
    assert plan.launches - l0 == 2, "one az_pack_f16_multi launch per step from the second step on"
    n_images = sum(1 for e in plan.entries.values() if e.wref() is not None and any(e.wref() is p for p in a.parameters()))
    assert n_images >= 80, n_images  # forward + input-gradient images of ~85 convolutions (some layers take other routes)
    monkeypatch.setattr(conv3d, "_PLAN_ON", False)
    rb = _run_steps(b, il, ir, gt, md, 3)
    for step, ((la, ga), (lb, gb)) in enumerate(zip(ra, rb)):
        # same bytes in the packed images => same arithmetic; what differs is the float-atomic order of the weight-gradient
        # flushes (first step: 2e-4 of relative L2, tests/test_gpu_overlap.py), amplified by Adam's sign-like first update
        assert abs(la - lb) <= (1e-5 if step == 0 else 2e-2) * abs(lb), (step, la, lb)
        if step == 0:
            for k in gb:
                d = (ga[k] - gb[k]).double().norm() / (gb[k].double().norm() + 1e-30)
                assert d <= 2e-4, (k, float(d))


def test_the_plan_does_not_answer_for_a_dead_parameter():
    plan = conv3d.PackPlan(DEV)
    w = torch.nn.Parameter(seeded((32, 32, 3, 3, 3), 9, -0.3, 0.3).to(DEV))
    plan.prepack([w])
    e, _ = plan.lookup(w, conv3d.PACK_3D_ROLL, 32, 32, 32, 32, 32 * 27, 27, 27, False)
    assert e is not None
    ptr = w.data_ptr()
    alias = w.detach()  # (same storage: what a gated weight of a training pass looks like)
    assert plan.lookup(alias, conv3d.PACK_3D_ROLL, 32, 32, 32, 32, 32 * 27, 27, 27, False)[0] is e
    del w, alias
    gc.collect()
    other = torch.empty(32 * 32 * 27, device=DEV)  # may well land on the freed address
    probe = other if other.data_ptr() == ptr else torch.empty(0, device=DEV)
    assert plan.lookup(probe, conv3d.PACK_3D_ROLL, 32, 32, 32, 32, 32 * 27, 27, 27, False)[0] is None
    plan.prepack([])
    assert not plan.entries and not plan.registered
