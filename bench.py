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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["hip", "miopen"], default=None,
                    help="3-D aggregation backend (miopen = PyTorch-eager A/B baseline)")
    ap.add_argument("--cpu-sample", choices=["full", "crop"], default="crop")
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="do not set torch.backends.cudnn.benchmark (the reference sets it, train.py:38)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="rehearsal only: 'gloo' lets several ranks share ONE GPU (with --single-device)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0")
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


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference's eager op sequence) timed on this
    box's host cores on a BOUNDED sample of the same workload: one pair, fwd+loss+bwd, on
    the reference's own training crop (256x512, configs/config.py:9-10) at the full D=192,
    scaled to the metric's 544x960 pairs by the pixel ratio (every stage of the path is
    linear in H*W).  A full-size pair takes > 6 min on 16 cores, too long for a default run
    (use --cpu-sample full to time it anyway)."""
    from oracle import psmnet_oracle as po

    # the GPU box exposes every host CPU (256) but one GPU's share is 16 cores; more
    # threads than that only oversubscribes the node (measured: 425 s instead of ~10 s)
    cores = int(os.environ.get("AZ_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
    torch.set_num_threads(cores)
    md = args.maxdisp
    hp = args.height + (-args.height) % 32
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
    preds = model(il, ir)
    loss = po.psmnet_disp_loss(preds, gt, po.disparity_mask(gt, md))
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * scale), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": sample, "sample_seconds": dt}


def main():
    args = parse()
    from activezero_amd import dist as azdist

    rank, local_rank, world = azdist.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    azdist.init(args.dist_backend)  # "nccl" = RCCL on ROCm

    from activezero_amd import agg3d, conv3d, profiler
    from activezero_amd.nets.psmnet.psmnet_3 import PSMNet

    if args.backend:
        agg3d.set_backend(args.backend)
    # the reference enables cudnn.benchmark (train.py:38); on ROCm this is MIOpen's
    # exhaustive find for the adjacent 2-D convolutions (paid once, during warm-up)
    torch.backends.cudnn.benchmark = not args.no_miopen_find
    torch.manual_seed(1)  # configs/config.py:100
    model = PSMNet(args.maxdisp).to(device).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999))
    net = azdist.wrap(model, device)
    il, ir, gt = synth_batch(args.batch, args.height, args.width, args.maxdisp, device,
                             azdist.rank_seed(1234, rank))

    def step():
        opt.zero_grad(set_to_none=True)
        loss = disp_loss(net(il, ir), gt, args.maxdisp)
        loss.backward()
        opt.step()
        return loss

    fence = azdist.fence

    def note(msg):  # progress on stderr (the JSON line is the only stdout output)
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"model ready, {args.warmup} warm-up steps")
    for _ in range(args.warmup):
        step()
    fence()
    note(f"timing {args.steps} steps")
    profiler.start()  # HIP events around the dominant kernel, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof = profiler.stop()
    dt = azdist.max_over_ranks(dt, device)

    if rank == 0:
        pairs = args.batch * world * args.steps
        out = {
            "metric": "stereo pairs/sec (540x960, D=192) fwd+bwd",
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: PSMNet {args.height}x{args.width} (padded to "
                                   f"{args.height + (-args.height) % 32} rows) D={args.maxdisp} fwd+bwd+Adam, "
                                   f"supervised disparity loss, batch {args.batch} per GPU",
                       "global_batch": args.batch * world, "height": args.height,
                       "width": args.width, "maxdisp": args.maxdisp,
                       "parallelism": f"dp{world}", "agg3d_backend": agg3d.BACKEND,
                       "arithmetic": "fp32 results; conv/deconv/dgrad on "
                                     + ("bf16x6 split MFMA" if conv3d.PRECISION else "fp32 MFMA")
                                     + ", wgrad on " + ("bf16x6 split MFMA" if conv3d.WGRAD_PRECISION else "fp32 MFMA")
                                     + ", fp32 accumulation everywhere"},
            "loss": float(loss.item()),
            "peak_mem_gb": torch.cuda.max_memory_allocated(device) / 2 ** 30,
            "roofline": profiler.roofline(prof, os.path.join(REPO, "profiles", "pmc_traffic_b4.json")),
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline and world == 1:
            note(f"{1e3 * dt / args.steps:.1f} ms/step; timing the CPU baseline sample ({args.cpu_sample})")
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    azdist.shutdown()


if __name__ == "__main__":
    main()
