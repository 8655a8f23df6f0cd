#!/usr/bin/env python3
"""Per-phase cycle sums of az_conv3d_s2roll.hip's waves (a -DS2R_STAMP build: tools/build_variant.sh s2rstamp
az_conv3d_s2roll.hip -DS2R_STAMP, run with AZ_LIB_PATH=.../libazhip_s2rstamp.so): where a workgroup's plane period goes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import _lib, conv3d
dev = torch.device("cuda:0")
lib = _lib.lib()
fn = lib.az_debug_s2roll_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B, q = 4, (48, 136, 240)
x = torch.randn(B, *q, 32, device=dev)
w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
for _ in range(3):
    conv3d._conv(x, w, conv3d.CONV_S2, conv3d.F16X3, stats=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
fn(buf, 1)
n = 5
for _ in range(n):
    conv3d._conv(x, w, conv3d.CONV_S2, conv3d.F16X3, stats=True)
torch.cuda.synchronize()
fn(buf, 0)
v = [int(b) for b in buf]
names = ["loads in flight", "split + LDS writes", "barrier behind staging", "taps", "barrier behind taps", "epilogue"]
planes = v[6]
print(f"wave-planes {planes / n:.0f} per launch; cycles per wave and plane (s_memtime ticks = 100 MHz? see total):")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {v[i] / planes:9.1f}")
print(f"  sum of phases              {sum(v[:6]) / planes:9.1f}   whole waves / planes {v[7] / planes:9.1f}")
