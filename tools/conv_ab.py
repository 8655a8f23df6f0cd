#!/usr/bin/env python3
"""A/B timing of conv kernel variants selected by environment (one process per variant,
interleaved by the caller).  Prints ms per launch for the V0 32->32 and V1 64->64 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d

def bench(shape, cin, cout, mode, reps=10):
    dev = torch.device("cuda:0")
    x = torch.randn(*shape, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05 if mode != 2 else torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    pk, ci, co = conv3d._pack_forward(w, mode, conv3d.DEFAULT_ARITH.conv)
    for _ in range(2):
        conv3d._run_gather(x, pk, mode, ci, co, conv3d.DEFAULT_ARITH.conv, stats=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        conv3d._run_gather(x, pk, mode, ci, co, conv3d.DEFAULT_ARITH.conv, stats=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

print("prec", os.environ.get("AZ_CONV_PRECISION"), "map", os.environ.get("AZ_CONV_MAP"),
      "V0 32->32 s1: %.3f ms" % bench((4, 48, 136, 240), 32, 32, 0),
      "| V0 64->32 s1: %.3f" % bench((4, 48, 136, 240), 64, 32, 0),
      "| V1 64->64 s1: %.3f" % bench((4, 24, 68, 120), 64, 64, 0),
      "| V0->V1 32->64 s2: %.3f" % bench((4, 48, 136, 240), 32, 64, 1),
      "| V1->V0 64->32 T2: %.3f" % bench((4, 24, 68, 120), 64, 32, 2))
