#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in one csrc/*.hip (cross-compiles, no GPU).

    python tools/kernel_resources.py az_conv3d.hip [filter] [-- extra hipcc flags]
"""
import os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = os.path.join(REPO, "activezero_amd", "csrc", args[0])
flt = args[1] if len(args) > 1 else ""
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(td, "x.o")] + extra,
                       capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
for b in r.stderr.split("Function Name: ")[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:70]:70s} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")
