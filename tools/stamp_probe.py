#!/usr/bin/env python3
"""Diagnostic: per-phase wave-cycle shares of the conv gather kernel (library built with
AZ_HIPCC_EXTRA=-DCV_STAMP).  The stamped build is for shares only, never for timing."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d, _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.az_debug_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = torch.randn(4, 48, 136, 240, 32, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
buf = (ctypes.c_ulonglong * 8)()
for prec in ("bf16x6", "fp32"):
    P = conv3d.Arith.of(prec).conv
    pk, ci, co = conv3d._pack_forward(w, 0, P)
    conv3d._run_gather(x, pk, 0, ci, co, P, stats=True); torch.cuda.synchronize()
    lib.az_debug_conv_stamps(buf, 1)
    reps = int(os.environ.get("STAMP_REPS", "600"))   # ~1.5 s of back-to-back launches: steady-state clock
    for _ in range(reps):
        conv3d._run_gather(x, pk, 0, ci, co, P, stats=True)
    torch.cuda.synchronize()
    lib.az_debug_conv_stamps(buf, 1)
    conv3d._run_gather(x, pk, 0, ci, co, P, stats=True); torch.cuda.synchronize()
    lib.az_debug_conv_stamps(buf, 1)
    n = buf[5]
    print(prec, "in-kernel clock %.3f GHz (shader cycles / 100 MHz ticks)" % (0.1 * buf[6] / max(buf[7], 1)))
    names = ["prologue(issue0)", "commit", "issue(next)", "taps(MFMA)", "epilogue"]
    tot = sum(buf[i] for i in range(5))
    print(prec, "waves", n, "cycles/wave", tot / n, {nm: round(buf[i] / n) for i, nm in enumerate(names)})
