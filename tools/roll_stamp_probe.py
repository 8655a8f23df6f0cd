#!/usr/bin/env python3
"""Phase breakdown of az_conv3d_roll.hip from the -DR16_STAMP diagnostic build (tools/build_variant.sh stamp
az_conv3d_roll.hip -DR16_STAMP; run with AZ_LIB_PATH=.../libazhip_stamp.so).  Shader cycles per wave and phase."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import _lib, conv3d
lib = _lib.lib()
lib.az_debug_roll_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
x = torch.randn(4, 48, 136, 240, 32, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
A = conv3d.DEFAULT_ARITH
pk, ci, co = conv3d._pack_forward(w, 0, A.conv)
pd = conv3d._pack(w, 32, 32, 27, 32 * 27, True, conv3d._layout(A.conv, 0, 32))
names = ["prologue", "stage bodies", "stage barriers", "tail", "-", "-", "-", "-", "kernel", "waves"]
for tag, fn in (("fwd+stats", lambda: conv3d._run_gather(x, pk, 0, ci, co, A.conv, stats=True)),
                ("dgrad", lambda: conv3d._run_gather(x, pd, 0, 32, 32, A.conv, tag="dgrad"))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 10)()
    lib.az_debug_roll_stamps(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    lib.az_debug_roll_stamps(buf, 1)
    waves = buf[9]
    print(f"{tag}: {a.elapsed_time(b) / 5:.3f} ms per launch, {waves // 5} waves")
    for i in range(9):
        print(f"   {names[i]:16s} {buf[i] / waves:12.0f} cycles/wave  {100.0 * buf[i] / buf[8]:5.1f} %")
