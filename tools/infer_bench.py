#!/usr/bin/env python3
"""Eval-mode forward latency (inference, under no_grad): BASELINE configs[0] (1 x 256x512, D=64) and the
full-size pair (540x960 padded to 544, D=192).  Eval uses the fused cost volume (never materialised)
and BatchNorm folded into the conv epilogues."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from activezero_amd.nets.psmnet.psmnet_3 import PSMNet  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
for b, h, w, md in ((1, 256, 512, 64), (1, 540, 960, 192), (4, 540, 960, 192)):
    torch.manual_seed(1)
    model = PSMNet(md).to(dev).eval()
    il, ir, _ = bench.synth_batch(b, h, w, md, dev, 7)
    with torch.no_grad():
        for _ in range(3):
            model(il, ir)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 10
        for _ in range(n):
            out = model(il, ir)
        e.record()
        torch.cuda.synchronize()
    ms = a.elapsed_time(e) / n
    print(f"eval forward B={b} {h}x{w} D={md}: {ms:.2f} ms/batch, {b / ms * 1e3:.1f} pairs/s, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GB")
