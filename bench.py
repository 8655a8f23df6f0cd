#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's metric:

    stereo pairs/sec (540x960, D=192) fwd+bwd     [configs[1]: batch 4 per GPU,
    PSMNet, supervised disparity loss only, Adam step included]

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; N>1 shards pairs across ranks (data parallel, RCCL gradient
all-reduce through DistributedDataParallel), weak scaling: batch 4 per GPU.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).

Synthetic data: rand images, ImageNet-normalised, 540 rows padded
to 544 on top as test.py does; smooth ground-truth disparity in (5,185); weights
from the reference init rule under manual_seed(1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

H_IMG, W_IMG, H_PAD, MAXDISP = 540, 960, 544, 192


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="pairs per GPU (config 2: 4)")
    ap.add_argument("--height", type=int, default=H_IMG)
    ap.add_argument("--width", type=int, default=W_IMG)
    ap.add_argument("--maxdisp", type=int, default=MAXDISP)
    ap.add_argument("--workload", choices=["supervised", "mixed", "raft"], default="supervised",
                    help="supervised = BASELINE configs[1] (the headline metric); mixed = configs[2]/[3]: the "
                         "default.yaml iteration (train.py:220-432): sim step (disparity + temporal-IR patch "
                         "reprojection loss) then real step (reprojection loss only), 6-channel PSMNet; "
                         "raft = configs[4]'s hot path: CorrBlock1D volume + pyramid at [B,256,H/4,W/4], then 22 x "
                         "(4-level lookup + ConvGRU update at the 1/4 resolution), forward + backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-steps", type=int, default=1,
                    help="timed steps of the same workload on stock PyTorch-ROCm operators (tools/eager_psmnet.py) "
                         "after the measurement, rank 0 at N=1 only: the 'PyTorch-eager' denominator; 0 = skip")
    ap.add_argument("--eager-batch", type=int, default=1, help="pairs per PyTorch-eager step")
    ap.add_argument("--eager-tuned", action="store_true",
                    help="PyTorch-eager legs with torch.backends.cudnn.benchmark = True as the reference sets it "
                         "(train.py:38): MIOpen's find phase runs in 2 untimed warm-up steps (minutes), then >= 3 timed "
                         "steps at --eager-batch.  The result of such a run is committed under profiles/ and quoted "
                         "by default runs as eager_gpu.tuned (carried_from)")
    ap.add_argument("--no-stage-bench", action="store_true",
                    help="skip the cost-volume + 3-D aggregation + soft-argmin stage comparison (eager_stage)")
    ap.add_argument("--cpu-sample", choices=["full", "crop"], default="full",
                    help="CPU baseline sample: one full-size pair (about 20 s on 16 cores) or the 256x512 crop scaled by the pixel ratio")
    ap.add_argument("--dist-backend", default="nccl",
                    help="rehearsal only: 'gloo' lets several ranks share ONE GPU (with --single-device)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N-rank launch path (rendezvous, assertions, sharded synthetic data, DDP "
                         "wrap, barrier + max-over-ranks timing, the JSON line) with a small stand-in model and "
                         "--dist-backend gloo: no GPU is touched, `value` is NOT a measurement (tests/test_bench_dryrun_cpu.py)")
    ap.add_argument("--no-solo-probe", action="store_true",
                    help="skip the stand-alone timing of the V0 launch shapes and the HBM copy probe after the timed region "
                         "(kernel traces of the step: tools/trace_step.sh)")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="A/B: weight-gradient kernels in order on the main stream (activezero_amd/overlap.py)")
    return ap.parse_args()


def synth_batch(b, h_img, w_img, maxdisp, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
    pad = (-h_img) % 32  # 540 -> 544 (hourglass needs H/4 divisible by 4; test.py:137-146)

    def image():
        x = (torch.rand(b, 3, h_img, w_img, device=device, generator=g) - mean) / std
        return F.pad(x, (0, 0, pad, 0)).contiguous()

    low = torch.randn(b, 1, (h_img + pad) // 16, w_img // 16, device=device, generator=g)
    low = F.avg_pool2d(F.pad(low, (1, 1, 1, 1), mode="replicate"), 3, 1)
    gt = 5.0 + (maxdisp - 12.0) * torch.sigmoid(
        F.interpolate(low, size=(h_img + pad, w_img), mode="bilinear", align_corners=False))
    gt[:, :, :pad] = 0.0  # padded rows carry no ground truth
    return image(), image(), gt.contiguous()


def disp_loss(preds, gt, maxdisp):
    """utils/losses.py:7-15 psmnet_disp with the train.py:272 mask (0 < gt < maxdisp): the fused K12
    kernel -- one pass, no boolean-index compaction, no host sync."""
    from activezero_amd.utils import disp_losses
    return disp_losses.psmnet_disp_range(preds, gt, maxdisp)


def synth_patterns(b, h_img, w_img, gt, device, seed):
    """Temporal-IR pattern pair: binary dots, Bernoulli(0.25) (datasets/dataset_utils.py:43-46);
    the right pattern is the left one moved by the ground-truth disparity with the scatter warp (K1), so
    that the reprojection loss has signal."""
    from activezero_amd.utils import warp_ops
    g = torch.Generator(device=device).manual_seed(seed)
    pad = (-h_img) % 32
    pl = (torch.rand(b, 1, h_img + pad, w_img, device=device, generator=g) < 0.25).float()
    pr = warp_ops.apply_disparity_cu(pl.contiguous(), (-gt).round().int().contiguous())  # left -> right: x - d
    return pl.contiguous(), pr.contiguous()


class heartbeat:
    """a line on stderr every minute while a long host-side leg runs (a silent process is taken to be hung)"""

    def __init__(self, what):
        self.what = what

    def __enter__(self):
        import threading
        self.stop = threading.Event()
        t0 = time.time()

        def run():
            while not self.stop.wait(60.0):
                print(f"[bench] {self.what}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        return False


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference's eager op sequence) timed on this box's host cores on
    BOUNDED samples of the workload.  Two legs, both reported:
      config1 -- BASELINE.json configs[0] exactly: one 256x512 pair, D=64, eval forward;
      value   -- the metric's unit: one pair, fwd+loss+bwd, on the reference's training crop (256x512,
                 configs/config.py:9-10) at the full D=192, scaled to 544x960 pairs by the pixel ratio
                 (every stage of the path is linear in H*W) with --cpu-sample crop; the default times one FULL-size
                 pair, unscaled (19.5 s on 16 cores, profiles/r03_bench_b4_cpu_full_sample.json)."""
    from oracle import psmnet_oracle as po

    # the GPU box exposes every host CPU (256) but one GPU's share is 16 cores; more
    # threads than that only oversubscribes the node (measured: 425 s instead of ~10 s)
    cores = int(os.environ.get("AZ_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
    torch.set_num_threads(cores)
    md = args.maxdisp
    hp = args.height + (-args.height) % 32
    torch.manual_seed(1)
    m1 = po.PSMNetOracle(64, 3).eval()
    il, ir, _ = synth_batch(1, 256, 512, 64, "cpu", 98)
    with torch.no_grad():
        m1(il, ir)  # warm the allocator / thread pool
        t0 = time.perf_counter()
        m1(il, ir)
        dt1 = time.perf_counter() - t0
    config1 = {"value": 1.0 / dt1, "unit": "pairs/s", "seconds": dt1,
               "sample": "BASELINE configs[0] exactly: 1 pair 256x512, D=64, eval forward, unscaled"}
    if args.cpu_sample == "full":
        h, w, scale = args.height, args.width, 1.0
        sample = f"1 pair {h}x{w} (padded to {hp}), D={md}, fwd+loss+bwd, 1 step, unscaled"
    else:
        h, w = 256, 512
        scale = (hp * args.width) / float(h * w)
        sample = (f"1 pair 256x512 crop, D={md}, fwd+loss+bwd, 1 step; seconds x {scale:.3f} "
                  f"(= {hp}x{args.width} / 256x512 pixels) -> pairs/s at full size")
    torch.manual_seed(1)
    model = po.PSMNetOracle(md, 3).train()
    il, ir, gt = synth_batch(1, h, w, md, "cpu", 99)
    t0 = time.perf_counter()
    with heartbeat("CPU baseline sample"):
        preds = model(il, ir)
        loss = po.psmnet_disp_loss(preds, gt, po.disparity_mask(gt, md))
        loss.backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * scale), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": sample, "sample_seconds": dt, "config1": config1}


TUNED_EAGER_JSON = os.path.join(REPO, "profiles", "r04_eager_tuned.json")
# MIOpen's user find-db of a `bench.py --eager-tuned` run on this image, committed (round 4): with it the eager legs of a
# DEFAULT run use the reference's own setting, cudnn.benchmark = True (train.py:38), LIVE -- the find phase reads its
# results from the db instead of timing every solver for minutes.  The directory is only ever given to MIOpen, i.e. to
# the eager legs; this library links no MIOpen.
MIOPEN_DB_DIR = os.path.join(REPO, "profiles", "miopen_db")


def _have_miopen_db():
    return os.path.isdir(MIOPEN_DB_DIR) and any(f.endswith(".txt") or f.endswith(".db") for f in os.listdir(MIOPEN_DB_DIR))


def _eager_mode(args):
    """(cudnn.benchmark, warm-up steps, timed steps, label)"""
    if args.eager_tuned:
        return True, 2, max(3, args.eager_steps), "cudnn.benchmark = True as train.py:38 sets it (MIOpen find phase in the warm-up steps)"
    if _have_miopen_db() and os.environ.get("AZ_BENCH_EAGER_UNTUNED") != "1":
        return True, 2, max(3, args.eager_steps), ("cudnn.benchmark = True as train.py:38 sets it; MIOpen's find results come from "
                                                    "the committed user find-db profiles/miopen_db (made by a --eager-tuned run on "
                                                    "this image), so the tuned solvers run LIVE in this run")
    return False, 1, args.eager_steps, ("UNTUNED MIOpen: cudnn.benchmark off (immediate-mode solvers; the find phase of the 3-D "
                                        "convolutions takes minutes) -- a lower bound of what the reference's configuration reaches; "
                                        "the tuned number is eager_gpu.tuned")


def _carried_tuned(key):
    """the committed result of a `bench.py --eager-tuned` run, labelled as carried"""
    if not os.path.exists(TUNED_EAGER_JSON):
        return None
    rec = json.load(open(TUNED_EAGER_JSON)).get(key)
    if rec is None:
        return None
    return dict(rec, carried_from=os.path.relpath(TUNED_EAGER_JSON, REPO))


def eager_gpu(args, model, il, ir, gt, device):
    """The same supervised step on stock PyTorch-ROCm operators over the same module (tools/eager_psmnet.py):
    the 'PyTorch-eager' leg.  A reported comparison, never `vs_baseline` (BASELINE.md publishes no number).
    Default: find mode off, ONE pair per step, one timed step (the run must finish within minutes);
    --eager-tuned: the reference's own setting, see _eager_mode."""
    from tools import eager_psmnet

    bench_flag, nwarm, nsteps, label = _eager_mode(args)
    torch.backends.cudnn.benchmark = bench_flag
    if args.eager_batch < il.shape[0]:
        il, ir, gt = (t[:args.eager_batch].contiguous() for t in (il, ir, gt))
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999))

    def step():
        opt.zero_grad(set_to_none=True)
        loss = eager_psmnet.eager_loss(eager_psmnet.eager_forward(model, il, ir), gt, args.maxdisp)
        loss.backward()
        opt.step()

    with heartbeat("PyTorch-eager warm-up"):
        for _ in range(nwarm):  # kernel selection / find phase, allocator
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nsteps
    torch.backends.cudnn.benchmark = False
    return {"value": il.shape[0] / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt, "steps": nsteps,
            "batch": il.shape[0], "warmup": nwarm, "tuned": bench_flag,
            "what": "same model/data/step through stock PyTorch-ROCm operators (MIOpen conv2d/conv3d, ATen "
                    "batch_norm/interpolate/softmax); " + label}


def eager_stage(args, model, device):
    """The stage the north-star target is worded on -- cost volume + 3-D aggregation + soft-argmin regression,
    forward + loss + backward, from the two [B,32,H/4,W/4] feature maps on (psmnet_3.py:149-220) -- through this
    library and through stock PyTorch-ROCm operators, same module, same inputs.  pairs/s each and their ratio."""
    from tools import eager_psmnet

    hp = args.height + (-args.height) % 32
    g = torch.Generator(device=device).manual_seed(77)
    b = args.batch
    feats = [torch.randn(b, 32, hp // 4, args.width // 4, device=device, generator=g).contiguous(memory_format=torch.channels_last)
             for _ in range(2)]
    _, _, gt = synth_batch(b, args.height, args.width, args.maxdisp, device, 78)

    def lib_step(fl, fr, gtb):
        fl, fr = fl.detach().requires_grad_(), fr.detach().requires_grad_()
        for p in model.parameters():
            p.grad = None
        loss = disp_loss(model._from_features(fl, fr, model._pass_arith(fl)), gtb, args.maxdisp)
        loss.backward()

    def eager_step(fl, fr, gtb):
        fl, fr = fl.detach().contiguous().requires_grad_(), fr.detach().contiguous().requires_grad_()
        for p in model.parameters():
            p.grad = None
        loss = eager_psmnet.eager_loss(eager_psmnet.eager_from_features(model, fl, fr, (hp, args.width)), gtb, args.maxdisp)
        loss.backward()

    def timeit(fn, argv, nwarm, nsteps):
        for _ in range(nwarm):
            fn(*argv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn(*argv)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nsteps

    dt_lib = timeit(lib_step, (feats[0], feats[1], gt), 1, 3)
    bench_flag, nwarm, nsteps, label = _eager_mode(args)
    torch.backends.cudnn.benchmark = bench_flag
    eb = min(args.eager_batch, b)
    with heartbeat("PyTorch-eager stage"):
        dt_eager = timeit(eager_step, (feats[0][:eb], feats[1][:eb], gt[:eb]), nwarm, max(1, nsteps))
    torch.backends.cudnn.benchmark = False
    lib, eag = b / dt_lib, eb / dt_eager
    # one level, scalars only
    return {"ratio": lib / eag, "unit": "pairs/s",
            "this_library_value": lib, "this_library_ms_per_step": 1e3 * dt_lib, "this_library_batch": b,
            "eager_value": eag, "eager_ms_per_step": 1e3 * dt_eager, "eager_batch": eb, "eager_tuned": bench_flag,
            "stage": "cost volume + 3-D aggregation (25 Conv3d/ConvTranspose3d + BatchNorm3d) + soft-argmin, fwd + loss + bwd "
                     "from the feature maps (psmnet_3.py:149-220)",
            "eager_what": label}


def solo_probe(args, device):
    """The V0 launch shapes (B x 48 x 136 x 240 voxels, 32 -> 32 channels: the kernels with the most time in the step)
    timed ALONE, after the timed region: HIP events over 10 launches behind 30 warm-up launches.  The in-step durations of
    the same scopes are those of kernels sharing the chip with the other stream."""
    from activezero_amd import conv3d
    hp = args.height + (-args.height) % 32
    shape = (args.batch, args.maxdisp // 4, hp // 4, args.width // 4, 32)
    g = torch.Generator(device=device).manual_seed(5)
    x = torch.randn(shape, device=device, generator=g)
    dy = torch.randn(shape, device=device, generator=g) * 1e-4
    w = torch.randn(32, 32, 3, 3, 3, device=device, generator=g) * 0.05
    A = conv3d.DEFAULT_ARITH
    bwd = conv3d.F16X3 if A.bwd16 else A.conv
    fns = {"conv3d_m0_32_32": lambda: conv3d._conv(x, w, conv3d.CONV_S1, A.conv, stats=True),
           "dgrad_m0_32_32": lambda: conv3d._input_grad(dy, w, conv3d.CONV_S1, 32, 32, bwd),
           "conv_wgrad_s1_32_32": lambda: conv3d._weight_grad(x, dy, conv3d.CONV_S1, 32, 32, conv3d.F16X3 if A.bwd16 else A.wgrad)}
    # BatchNorm backward of a V0 tensor (reduce + apply, ReLU mask recomputed from raw, max |dx| taken): scope bn3d_bwd_32
    from activezero_amd import _lib
    from activezero_amd.ops import _call, _p, _stream
    nv = x.numel() // 32
    wsb = _lib.lib().az_bn3d_bwd_workspace(nv, 32)
    ws, dx = torch.empty(wsb // 4, device=device), torch.empty_like(x)
    cvec = [torch.ones(32, device=device) for _ in range(3)] + [torch.zeros(32, device=device)]
    small = [torch.empty(32, device=device), torch.empty(32, device=device), torch.empty(32, 3, device=device), torch.zeros(1024, device=device)]
    fns["bn3d_bwd_32"] = lambda: _call("az_bn3d_bwd", _p(dx), None, _p(small[0]), _p(small[1]), _p(small[2]), _p(ws), wsb, _p(dy), None,
                                       _p(x), _p(cvec[3]), _p(cvec[0]), _p(cvec[1]), _p(cvec[2]), _p(cvec[3]), 1, nv, 32, _p(small[3]), int(conv3d.PRESPLIT), _stream())
    with torch.no_grad():
        for _ in range(10):
            for f in fns.values():
                f()
        out = {}
        for name, f in fns.items():
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                f()
            b.record()
            torch.cuda.synchronize()
            out[name] = a.elapsed_time(b) / 10
    return out


def hbm_probe(device):
    """Measured device-to-device copy rate on this box (the second denominator beside the 8 TB/s spec
    figure): a 1 GiB fp32 tensor through this library's float4 grid-stride copy kernel (az_hbm_copy_probe: the stream
    MI355X_MICROARCH.md quotes 6.29 TB/s for), read + write bytes over the median of 7 runs after a warm-up."""
    from activezero_amd.ops import _call, _p, _stream
    n = 1 << 28
    x = torch.empty(n, dtype=torch.float32, device=device).normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        _call("az_hbm_copy_probe", _p(y), _p(x), n, _stream())
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _call("az_hbm_copy_probe", _p(y), _p(x), n, _stream())
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    return {"kind": "float4 grid-stride copy kernel (az_hbm_copy_probe), 1 GiB fp32, read + write",
            "GB/s": 2.0 * n * 4 / 1e9 / (ms * 1e-3), "spec_GB/s": 8000.0, "guide_GB/s": 6290.0}


def raft_workload(args, rank, local_rank, world):
    """BASELINE.json configs[4], the part of RAFT-Stereo that is on this path: the all-pairs
    1-D correlation volume and its 4-level pyramid (nets/raft/corr.py:115-161) built ONCE from two [B,256,H/4,W/4]
    feature maps, then raft_stereo.py:138-172's loop, 22 iterations of {lookup at the current coordinates, ConvGRU
    update of the 1/4-resolution hidden state (update.py:19-41, hidden 128, input 256)}, forward + backward to the
    feature maps and the GRU parameters.  The motion encoder / flow head (ordinary torch code, out of scope) are
    replaced by fixed tensors: the 36 lookup channels enter the GRU input beside 220 constant context channels,
    the coordinate update is the mean of the new state (detached, as RAFT detaches coords1)."""
    from activezero_amd import dist as azdist, profiler
    from activezero_amd.nets.raft.corr import CorrBlock1D
    from activezero_amd.nets.raft.gru import ConvGRU

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(0 if args.single_device else local_rank)
    device = torch.device("cuda", 0 if args.single_device else local_rank)
    azdist.init(args.dist_backend)
    hp = args.height + (-args.height) % 32
    b, h, w, iters = args.batch, hp // 4, args.width // 4, 22
    torch.manual_seed(1)
    gru = ConvGRU(128, 256).to(device)
    net = azdist.wrap(gru, device)
    opt = torch.optim.Adam(gru.parameters(), lr=2e-4)
    g = torch.Generator(device=device).manual_seed(azdist.rank_seed(4321, rank))
    rnd = lambda *shape: torch.randn(*shape, device=device, generator=g)
    f1, f2 = rnd(b, 256, h, w).requires_grad_(), rnd(b, 256, h, w).requires_grad_()
    h0 = torch.tanh(rnd(b, 128, h, w))
    cz, cr, cq = (0.5 * rnd(b, 128, h, w) for _ in range(3))
    ctx = rnd(b, 220, h, w)
    xs = torch.arange(w, device=device, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.arange(h, device=device, dtype=torch.float32).view(1, 1, h, 1).expand(b, 1, h, w)
    coords0 = torch.cat([xs, ys], 1).contiguous()

    def step():
        opt.zero_grad(set_to_none=True)
        f1.grad = f2.grad = None
        corr_fn = CorrBlock1D(f1, f2, radius=4, num_levels=4)
        state, coords, loss = h0, coords0 - 8.0, 0.0
        for it in range(iters):
            coords = coords.detach()
            corr = corr_fn(coords)                                  # [B, 36, h, w]
            state = net(state, cz, cr, cq, corr, ctx)
            loss = loss + (0.9 ** (iters - 1 - it)) * state.abs().mean()   # sequence_loss-style weights
            delta = state.mean(1, keepdim=True).detach()
            coords = torch.cat([coords[:, :1] + delta, coords[:, 1:]], 1)  # stereo: x only
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    azdist.fence()
    profiler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    azdist.fence()
    dt = time.perf_counter() - t0
    prof = profiler.stop()
    dt = azdist.max_over_ranks(dt, device)
    if rank == 0:
        vol = prof.get("corr1d_volume")
        roof = None
        if vol:
            # K10: 2 B H W1 W2 C flop; HBM: both feature maps in, the volume out
            nbytes = 4.0 * b * h * (2 * 256 * w + w * w)
            roof = {"kernel": "corr1d_volume (az_corr1d.hip)", "bound": "hbm", "achieved": nbytes / 1e9 / (vol["avg_ms"] * 1e-3),
                    "peak": 8000.0, "unit": "GB/s", "frac": nbytes / 1e9 / (vol["avg_ms"] * 1e-3) / 8000.0, "traffic": None,
                    "avg_launch_ms": vol["avg_ms"], "per_launch_work": nbytes,
                    "flops_per_launch": vol["flops"], "TFLOP/s": vol["flops"] / 1e12 / (vol["avg_ms"] * 1e-3),
                    "note": "41 flop per byte moved: on the bf16x6 pipe (416.7 TFLOP/s nominal) the contraction itself takes 0.04 ms, "
                            "the 393 MB of operands and volume 0.05 ms at 8 TB/s -- priced against HBM; others: every scope of the step",
                    "others": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"]} for k, v in prof.items()
                               if k != "corr1d_volume"}}
        print(json.dumps({
            "metric": "stereo pairs/sec (540x960) fwd+bwd, RAFT-Stereo correlation volume + 22 lookup/ConvGRU iterations",
            "value": b * world * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4] hot path: CorrBlock1D on [B,256,{h},{w}] feature maps (volume + 4-level pyramid), "
                                   f"{iters} x (lookup r=4 + ConvGRU(128, 256) update), fwd + loss + bwd + Adam, batch {b} per GPU; "
                                   "motion encoder / flow head / extractors not included (out of scope)",
                       "global_batch": b * world, "height": args.height, "width": args.width, "parallelism": f"dp{world}",
                       "dist_backend": (args.dist_backend if world > 1 else None)},
            "loss": float(loss.item()), "peak_mem_gb": torch.cuda.max_memory_allocated(device) / 2 ** 30,
            "roofline": roof, "cpu_baseline": None}), flush=True)
    azdist.shutdown()


def dry_run(args, rank, world):
    """The launch path of an N-rank run without a GPU: everything bench.py does around the step -- rendezvous from
    the torch.distributed.run environment, the world-size / backend assertions, per-rank synthetic data, the DDP
    wrap of activezero_amd.dist (one bucket), barrier + max-over-ranks timing, rank 0's JSON line -- with a small
    Conv3d + BatchNorm3d stand-in for PSMNet (the HIP kernels have no CPU fallback)."""
    import torch.distributed as dist
    from activezero_amd import dist as azdist

    if args.dist_backend != "gloo":
        raise SystemExit("--dry-run is a CPU rehearsal: pass --dist-backend gloo")
    azdist.init("gloo")
    if world > 1:
        assert dist.get_world_size() == args.gpus and dist.get_backend() == "gloo"
    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Conv3d(2, 8, 3, padding=1, bias=False), torch.nn.BatchNorm3d(8), torch.nn.ReLU(),
                                torch.nn.Conv3d(8, 1, 3, padding=1, bias=False))
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    net = azdist.wrap(model, torch.device("cpu"))
    g = torch.Generator().manual_seed(azdist.rank_seed(1234, rank))
    x = torch.randn(args.batch, 2, 6, 8, 12, generator=g)
    y = torch.randn(args.batch, 1, 6, 8, 12, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    azdist.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    azdist.fence()
    dt = azdist.max_over_ranks(time.perf_counter() - t0)
    # every rank drew its own pairs and holds the same parameters after the all-reduced steps
    digest = torch.cat([p.detach().flatten() for p in model.parameters()]).double().sum()
    lo, hi = digest.clone(), digest.clone()
    if world > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    first = x[0, 0, 0, 0, 0].double().clone()
    fmin, fmax = first.clone(), first.clone()
    if world > 1:
        dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(fmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            "metric": "stereo pairs/sec (540x960, D=192) fwd+bwd", "value": args.batch * world * args.steps / dt,
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dry_run": "CPU rehearsal of the launch / rendezvous / sharding / timing path with a stand-in model: "
                       "`value` is NOT a measurement",
            "config": {"workload": "dry run (stand-in Conv3d + BatchNorm3d model on CPU)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "dist_backend": args.dist_backend if world > 1 else None},
            "replicas_in_sync": bool(abs(hi.item() - lo.item()) <= 1e-9 * max(1.0, abs(hi.item()))),
            "ranks_drew_distinct_data": bool(world == 1 or fmax.item() != fmin.item()),
            "loss": float(loss.item())}), flush=True)
    azdist.shutdown()


def main():
    args = parse()
    from activezero_amd import dist as azdist

    rank, local_rank, world = azdist.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.workload == "raft":
        return raft_workload(args, rank, local_rank, world)
    if _have_miopen_db() or args.eager_tuned:
        # (before anything initialises MIOpen; read by the eager legs only)
        os.environ.setdefault("MIOPEN_USER_DB_PATH", MIOPEN_DB_DIR if _have_miopen_db() else os.path.join(REPO, "gpurun_out", "miopen_db"))
        os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
        if not args.eager_tuned and os.environ.get("AZ_BENCH_EAGER_UNTUNED") != "1":
            args.eager_batch = args.batch  # the db holds the shapes of the full batch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    azdist.init(args.dist_backend)  # "nccl" = RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        # self-describing multi-GPU runs: one rank per GPU over RCCL unless a rehearsal was asked for
        assert dist.get_world_size() == args.gpus
        assert dist.get_backend() == args.dist_backend, (dist.get_backend(), args.dist_backend)
        if not args.single_device:
            assert args.dist_backend == "nccl", "multi-GPU measurements run over RCCL (backend 'nccl')"
            assert torch.cuda.device_count() >= world, "one GPU per rank"

    from activezero_amd import profiler
    from activezero_amd.utils import reprojection

    mixed = args.workload == "mixed"
    if mixed:
        from activezero_amd.nets.psmnet.psmnet import PSMNet
    else:
        from activezero_amd.nets.psmnet.psmnet_3 import PSMNet
    torch.manual_seed(1)  # configs/config.py:100
    model = PSMNet(args.maxdisp).to(device).train()
    model.set_weight_grad_overlap(not args.no_wgrad_overlap)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999))
    net = azdist.wrap(model, device)
    seed = azdist.rank_seed(1234, rank)
    il, ir, gt = synth_batch(args.batch, args.height, args.width, args.maxdisp, device, seed)

    if not mixed:
        def step():
            opt.zero_grad(set_to_none=True)
            loss = disp_loss(net(il, ir), gt, args.maxdisp)
            loss.backward()
            opt.step()
            return loss
    else:
        # default.yaml iteration (train.py:220-432, utils/losses.py:81-156): ADAPTER=True -> 6-channel PSMNet
        # (the adapter itself is outside the path: its outputs are synthetic tanh(randn) images);
        # sim: psmnet_disp + 1.0 * patch reprojection (ps = 11, masked) on the sim IR patterns;
        # real: 1.0 * patch reprojection (no mask) on the real pair; each with its own backward + Adam step.
        g = torch.Generator(device=device).manual_seed(seed + 7)
        tl, tr = (torch.tanh(torch.randn(il.shape, device=device, generator=g)) for _ in range(2))
        rl, rr, rgt = synth_batch(args.batch, args.height, args.width, args.maxdisp, device, seed + 1000)
        rtl, rtr = (torch.tanh(torch.randn(il.shape, device=device, generator=g)) for _ in range(2))
        spl, spr = synth_patterns(args.batch, args.height, args.width, gt, device, seed + 11)
        rpl, rpr = synth_patterns(args.batch, args.height, args.width, rgt, device, seed + 12)
        mask = (gt < args.maxdisp) & (gt > 0)  # train.py:272
        ps = 11  # configs/config.py:41

        def step():
            opt.zero_grad(set_to_none=True)
            out = net(il, ir, tl, tr)
            loss = disp_loss(out, gt, args.maxdisp)
            loss = loss + reprojection.get_reproj_error_patch(spl, spr, out[0], mask, ps)[0]
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            out = net(rl, rr, rtl, rtr)
            real = reprojection.get_reproj_error_patch(rpl, rpr, out[0], None, ps)[0]
            real.backward()
            opt.step()
            return loss + real.detach()

    fence = azdist.fence

    def note(msg):  # progress on stderr (the JSON line is the only stdout output)
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"model ready, {args.warmup} warm-up steps")
    import contextlib
    # AZ_BENCH_HP=1 (scheduling experiment, tools/sched_ab.sh): the whole step on a HIGH-priority stream, so that the
    # main stream's short kernels are dispatched ahead of the side stream's pending weight-gradient workgroups
    hp_stream = torch.cuda.Stream(device=device, priority=-1) if os.environ.get("AZ_BENCH_HP") == "1" else None
    on_stream = (lambda: torch.cuda.stream(hp_stream)) if hp_stream is not None else contextlib.nullcontext
    with on_stream():
        for _ in range(args.warmup):
            step()
    fence()
    note(f"timing {args.steps} steps")
    profiler.start()  # HIP events around the dominant kernel, on the launch stream
    mem0 = torch.cuda.memory_stats(device)
    t0 = time.perf_counter()
    with on_stream():
        for _ in range(args.steps):
            loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof = profiler.stop()
    dt = azdist.max_over_ranks(dt, device)

    if rank == 0:
        pairs_per_step = args.batch * (2 if mixed else 1)  # mixed: one sim pair + one real pair per sample slot
        pairs = pairs_per_step * world * args.steps
        hp = args.height + (-args.height) % 32
        cname, wname = model.arith.names
        if mixed:
            workload = (f"configs[2]: default.yaml mixed-domain iteration, PSMNet(6-ch) {args.height}x{args.width} "
                        f"(padded to {hp} rows) D={args.maxdisp}: sim step (disparity + temporal-IR patch "
                        f"reprojection loss, ps=11) + real step (reprojection loss), 2 x (fwd+bwd+Adam), "
                        f"batch {args.batch} per GPU and domain")
            metric = "stereo pairs/sec (540x960, D=192) fwd+bwd, mixed-domain iteration (sim + real pairs)"
        else:
            workload = (f"configs[1]: PSMNet {args.height}x{args.width} (padded to {hp} rows) D={args.maxdisp} "
                        f"fwd+bwd+Adam, supervised disparity loss, batch {args.batch} per GPU")
            metric = "stereo pairs/sec (540x960, D=192) fwd+bwd"
        out = {
            "metric": metric,
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": args.batch * world, "height": args.height,
                       "width": args.width, "maxdisp": args.maxdisp,
                       "parallelism": f"dp{world}",
                       "dist_backend": (args.dist_backend + " (RCCL)" if args.dist_backend == "nccl" else args.dist_backend) if world > 1 else None,
                       "arithmetic": f"fp32 results; forward MFMA arithmetic {cname}, input / weight gradients {wname} "
                                     "(f16x3 = operands scaled by a power of two from the tensor's max |.| and split into "
                                     "two fp16 parts, three MFMAs per product, where a kernel for the shape exists; "
                                     "bf16x6 = exact 3-way bf16 split, six MFMAs per product), fp32 accumulation "
                                     "everywhere; every convolution on this library's kernels"},
            "loss": float(loss.item()),
            "peak_mem_gb": torch.cuda.max_memory_allocated(device) / 2 ** 30,
            "peak_reserved_gb": torch.cuda.max_memory_reserved(device) / 2 ** 30,
            # hipMalloc / hipFree calls of the caching allocator INSIDE the timed steps (a steady state has none)
            "allocator_in_timed_steps": {k: int(torch.cuda.memory_stats(device).get(k, 0) - mem0.get(k, 0))
                                         for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")},
            "roofline": profiler.roofline(prof, os.path.join(REPO, "profiles", "r05_pmc_traffic_b4.json"),
                                          os.path.join(REPO, "profiles", "r05_pmc_clock_b4.json"),
                                          ms_per_step=1e3 * dt / args.steps, steps=args.steps,
                                          solo=solo_probe(args, device) if (world == 1 and not mixed and not args.no_solo_probe) else None),
            "cpu_baseline": None,
            "eager_gpu": None,
            "eager_stage": None,
            "vs_baseline_basis": "null: BASELINE.md publishes no number for this metric; the PyTorch-eager legs "
                                 "(eager_gpu, eager_stage) are reported comparisons, not the baseline",
        }
        if out["roofline"] is not None and not args.no_solo_probe:
            rf = out["roofline"]
            detail = rf.pop("detail")  # (nested records stay last)
            detail["measured_hbm"] = hbm_probe(device)
            rf["measured_hbm_gbps"] = detail["measured_hbm"]["GB/s"]
            rf["detail"] = detail
        if world == 1 and not mixed and args.eager_steps > 0:
            note(f"{1e3 * dt / args.steps:.1f} ms/step; timing the PyTorch-eager step on the GPU")
            del opt
            if not args.no_stage_bench:
                out["eager_stage"] = eager_stage(args, model, device)
                tuned = _carried_tuned("eager_stage")
                if tuned is not None and not args.eager_tuned:
                    # the committed record of the --eager-tuned run that wrote the find-db: its eager time next to this run's
                    ce = tuned.get("eager_value", (tuned.get("eager") or {}).get("value"))
                    out["eager_stage"].update({"carried_eager_value": ce, "carried_from": tuned.get("carried_from"),
                                               "ratio_live_over_carried_eager": out["eager_stage"]["this_library_value"] / ce if ce else None})
            out["eager_gpu"] = eager_gpu(args, model, il, ir, gt, device)
            out["eager_gpu"]["x"] = out["value"] / out["eager_gpu"]["value"]
            tuned_run = _carried_tuned("eager_gpu")
            if tuned_run is not None and not args.eager_tuned:
                out["eager_gpu"].update({"carried_tuned_value": tuned_run["value"], "carried_tuned_ms_per_step": tuned_run.get("ms_per_step"),
                                         "carried_from": tuned_run.get("carried_from"),
                                         "x_live_over_carried_eager": out["value"] / tuned_run["value"]})
            if args.eager_tuned:  # the record a later default run quotes
                os.makedirs(os.path.dirname(TUNED_EAGER_JSON), exist_ok=True)
                json.dump({"eager_gpu": out["eager_gpu"], "eager_stage": out["eager_stage"],
                           "this_library_pairs_per_s": out["value"], "command": " ".join(sys.argv)},
                          open(os.path.join(REPO, "gpurun_out", "r04_eager_tuned.json")
                               if os.path.isdir(os.path.join(REPO, "gpurun_out")) else TUNED_EAGER_JSON, "w"), indent=1)
        if not args.no_cpu_baseline and world == 1:
            note("timing the CPU baseline samples")
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    azdist.shutdown()


if __name__ == "__main__":
    main()
