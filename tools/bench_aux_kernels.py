#!/usr/bin/env python3
"""Microbenchmarks of the bandwidth-class kernels at BASELINE.json's full size
(B=4, 544x960, D=192): time per launch (HIP events on the launch stream) against the
ALGORITHMIC bytes of SURVEY.md 8(d).  Prints a markdown table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import ops, conv3d
from activezero_amd.utils import reprojection as rp, warp_ops
from activezero_amd.nets.raft.corr import CorrBlock1D

dev = torch.device("cuda:0")
B, H, W, D = 4, 544, 960, 192
h, w, d, C = H // 4, W // 4, D // 4, 32


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rows = []
def row(name, ms, nbytes, note=""):
    rows.append(f"| {name} | {ms:.3f} | {nbytes / 1e6:.1f} | {nbytes / ms / 1e6:.0f} | {nbytes / ms / 1e6 / 8000 * 100:.0f} % | {note} |")

g = torch.Generator(device=dev).manual_seed(0)
# K1 scatter warp (C=1)
x = (40 * torch.rand(B, 1, H, W, device=dev, generator=g)).contiguous()
di = x.int()
row("K1 `az_warp_scatter` (C=1)", timeit(lambda: ops.warp_scatter(x, di, 1)), 4.0 * B * H * W * 3)
# K3 cost volume
fl, fr = torch.randn(B, C, h, w, device=dev), torch.randn(B, C, h, w, device=dev)
vol_bytes = 4.0 * B * (2 * C * h * w + 2 * C * d * h * w)
vol = ops.cost_volume(fl, fr, d)
row("K3 `az_cost_volume_fwd` (NCDHW)", timeit(lambda: ops.cost_volume(fl, fr, d)), vol_bytes)
gl, gr = torch.empty_like(fl), torch.empty_like(fr)
row("K3 `az_cost_volume_bwd` (NCDHW)", timeit(lambda: ops._call("az_cost_volume_bwd", gl.data_ptr(), gr.data_ptr(), vol.data_ptr(), B, C, d, h, w, ops._stream())), vol_bytes)
del vol
fln, frn = fl.permute(0, 2, 3, 1).contiguous(), fr.permute(0, 2, 3, 1).contiguous()
volc = ops.cost_volume_ndhwc(fln, frn, d)
row("K3 `az_cost_volume_fwd_ndhwc`", timeit(lambda: ops.cost_volume_ndhwc(fln, frn, d)), vol_bytes)
row("K3 `az_cost_volume_bwd_ndhwc`", timeit(lambda: ops._call("az_cost_volume_bwd_ndhwc", gl.data_ptr(), gr.data_ptr(), volc.data_ptr(), B, C, d, h, w, ops._stream())), vol_bytes)
del volc
# K6 soft-argmin
lg = (3 * torch.randn(B, 1, d, h, w, device=dev)).requires_grad_()
out = ops.softargmin(lg)
row("K6 `az_softargmin_fwd` (per head)", timeit(lambda: ops.softargmin(lg.detach())), 4.0 * B * (d * h * w + H * W), "VALU-issue bound: plane values in registers, one v_exp_f32 per disparity")
go = torch.randn_like(out)
glg = torch.empty_like(lg)
row("K6 `az_softargmin_bwd` (per head)", timeit(lambda: ops._call("az_softargmin_bwd", glg.data_ptr(), go.data_ptr(), lg.data_ptr(), None, None, B, d, h, w, ops._stream())), 4.0 * B * (2 * d * h * w + H * W), "recompute; per-wave private LDS gradient tiles, global float atomics")
# K7 gather warp
img = torch.randn(B, 1, H, W, device=dev)
dsp = (8 * torch.rand(B, 1, H, W, device=dev)).requires_grad_()
row("K7 `az_warp_gather_fwd` (C=1)", timeit(lambda: ops.warp_gather(img, dsp.detach())), 4.0 * B * H * W * 3)
# K8 patch reprojection ps=11
pl = (torch.rand(B, 1, H, W, device=dev) < 0.25).float()
pr = pl.roll(5, 3).contiguous()
mask = torch.rand(B, 1, H, W, device=dev) < 0.9
def k8f():
    return ops.patch_reprojection(pl, pr, dsp.detach(), mask, 11, want_vis=False)
row("K8 `az_patch_reproj_fwd` (ps=11)", timeit(k8f), 4.0 * B * H * W * 3.25, "band of rows at full width in LDS, four pixels per thread")
loss, _, _ = ops.patch_reprojection(pl, pr, dsp, mask, 11, want_vis=False)
def k8b():
    dsp.grad = None
    l, _, _ = ops.patch_reprojection(pl, pr, dsp, mask, 11, want_vis=False)
    l.backward()
tb = timeit(k8b)
row("K8 fwd+bwd (ps=11)", tb, 4.0 * B * H * W * (3.25 + 4.25))
row("K8 `az_patch_reproj_vis` (Fold image)", timeit(lambda: ops.patch_reprojection(pl, pr, dsp.detach(), mask, 11, want_vis=True)) - timeit(k8f), 4.0 * B * H * W * 3, "121 x 4 gathers per pixel; TensorBoard only")
# K9 LCN
im = torch.rand(B, 1, 540, 960, device=dev)
row("K9 `az_lcn` (k=11)", timeit(lambda: ops.local_contrast_norm(im, 11)), 4.0 * B * 540 * 960 * 3)
# BN passes on a V0 32-channel tensor
xv = torch.randn(B, d, h, w, C, device=dev)
yv = torch.empty_like(xv)
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
row("`az_bn3d_apply` (V0, 32 ch)", timeit(lambda: ops._call("az_bn3d_apply", yv.data_ptr(), xv.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, xv.numel() // C, C, None, ops._stream())), 8.0 * xv.numel())
# BatchNorm backward of the same tensor (five tensor passes: 2 x (dy, raw) in, dx out) with max |dx| taken on the way
from activezero_amd import _lib  # noqa: E402
gy, raw = torch.randn_like(xv), torch.randn_like(xv)
nv = xv.numel() // C
mean, invstd, gamma = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
dgm, dbt, coef, dxa = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, 3, device=dev), torch.zeros(1024, device=dev)
wsb = _lib.lib().az_bn3d_bwd_workspace(nv, C)
wsp = torch.empty(wsb // 4, device=dev)
row("`az_bn3d_bwd` (V0, 32 ch, ReLU mask recomputed, amax out)", timeit(lambda: ops._call(
    "az_bn3d_bwd", yv.data_ptr(), None, dgm.data_ptr(), dbt.data_ptr(), coef.data_ptr(), wsp.data_ptr(), wsb, gy.data_ptr(), None,
    raw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, nv, C, dxa.data_ptr(), 0,
    ops._stream())), 20.0 * xv.numel(), "reduce + apply (partials merged in the apply prologue)")
row("`az_bn3d_bwd` (V0, 32 ch), dx written pre-split", timeit(lambda: ops._call(
    "az_bn3d_bwd", yv.data_ptr(), None, dgm.data_ptr(), dbt.data_ptr(), coef.data_ptr(), wsp.data_ptr(), wsb, gy.data_ptr(), None,
    raw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, nv, C, dxa.data_ptr(), 1,
    ops._stream())), 20.0 * xv.numel(), "round 5: per-channel maxima in the reduce pass, bound + two fp16 parts in the apply pass")
row("`az_absmax` (V0, 32 ch)", timeit(lambda: ops._call("az_absmax", dxa.data_ptr(), xv.data_ptr(), xv.numel(), ops._stream())), 4.0 * xv.numel(), "the stand-alone amax pass (f16x3 operand scale) where no producer kernel took it")
import bench as _bench  # the SAME probe bench.py's roofline.measured_hbm_gbps comes from (one implementation, one size: VERDICT r4 item 9)
_hb = _bench.hbm_probe(dev)
row("`az_hbm_copy_probe` (float4 copy, 1 GiB in + 1 GiB out; bench.hbm_probe)", 2.0 * (1 << 30) / 1e6 / _hb["GB/s"], 2.0 * (1 << 30), "bench.py's measured_hbm kernel, median of 7 after 3 warm-up launches")
row("`az_hbm_copy_probe` on the V0 tensor above (0.8 GB in + 0.8 GB out), 10 launches behind 2", timeit(lambda: ops._call("az_hbm_copy_probe", yv.data_ptr(), xv.data_ptr(), xv.numel(), ops._stream())), 8.0 * xv.numel(), "the round-4 table's row: same kernel, this tool's generic timer")
# 32 -> 1 classifier convolution (VALU kernels)
wc = torch.randn(1, C, 3, 3, 3, device=dev) * 0.05
lo = torch.empty(B, d, h, w, device=dev); gxc = torch.empty_like(xv); gwc = torch.empty_like(wc)
row("K4c `az_conv3d_c1_fwd`", timeit(lambda: ops._call("az_conv3d_c1_fwd", lo.data_ptr(), xv.data_ptr(), wc.data_ptr(), None, None, None, B, d, h, w, ops._stream())), 4.0 * (xv.numel() + lo.numel()))
row("K4c `az_conv3d_c1_dgrad`", timeit(lambda: ops._call("az_conv3d_c1_dgrad", gxc.data_ptr(), lo.data_ptr(), wc.data_ptr(), B, d, h, w, ops._stream())), 4.0 * (xv.numel() + lo.numel()))
row("K4c `az_conv3d_c1_wgrad`", timeit(lambda: ops._call("az_conv3d_c1_wgrad", gwc.data_ptr(), xv.data_ptr(), lo.data_ptr(), None, None, B, d, h, w, ops._stream())), 4.0 * (xv.numel() + lo.numel()))
# K10/K11 RAFT
f1, f2 = torch.randn(B, 256, h, w, device=dev), torch.randn(B, 256, h, w, device=dev)
ms = timeit(lambda: CorrBlock1D(f1, f2), reps=5)
rows.append(f"| K10 `az_corr1d_volume` + 4 pools | {ms:.3f} | {2.0 * 256 * B * h * w * w / 1e9:.1f} GFLOP | {2.0 * 256 * B * h * w * w / ms / 1e9:.1f} TFLOP/s | {2.0 * 256 * B * h * w * w / ms / 1e9 / 157.3 * 100:.0f} % of fp32 MFMA | generic strided batched GEMM |")
blk = CorrBlock1D(f1, f2)
coords = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")[::-1], 0).float()[None].repeat(B, 1, 1, 1).to(dev)
coords[:, 0] -= 20
row("K11 lookup, 4 levels x 9 taps", timeit(lambda: blk(coords)), 4.0 * B * h * w * (36 + 1 + 2 * 36))
print("| kernel | ms/launch | algorithmic MB | GB/s | of 8 TB/s | note |\n|---|---|---|---|---|---|")
print("\n".join(rows))
