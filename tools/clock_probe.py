#!/usr/bin/env python3
"""Diagnostic: clock the chip holds inside the 8x16-patch conv kernel (library built with -DCV_STAMP,
path in AZ_LIB_PATH): shader cycles / 100 MHz real-time ticks after ~2 s of back-to-back launches."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import conv3d, _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.az_debug_m128_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = torch.randn(4, 48, 136, 240, 32, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk, ci, co = conv3d._pack_forward(w, 0, conv3d.DEFAULT_ARITH.conv)
buf = (ctypes.c_ulonglong * 4)()
reps = int(os.environ.get("STAMP_REPS", "1000"))
for _ in range(reps):
    conv3d._run_gather(x, pk, 0, ci, co, conv3d.DEFAULT_ARITH.conv, stats=True)
torch.cuda.synchronize()
lib.az_debug_m128_stamps(buf, 1)
t0 = time.perf_counter()
for _ in range(20):
    conv3d._run_gather(x, pk, 0, ci, co, conv3d.DEFAULT_ARITH.conv, stats=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
lib.az_debug_m128_stamps(buf, 1)
print("m128 32->32: %.3f ms/launch, in-kernel clock %.3f GHz, %.0f cycles/wave (%d waves)"
      % (dt * 1e3, 0.1 * buf[0] / max(buf[1], 1), buf[0] / max(buf[2], 1), buf[2] // 20))
