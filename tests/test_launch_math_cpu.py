"""The host-side launch arithmetic of the tiled kernels (activezero_amd/csrc/az_launch_math.h: XCD block map, depth /
image segments, work-item decode, slab staging indices, persistent-workgroup counts, the 32-bit offset guard) compiled
for the CPU with AddressSanitizer + UBSan and swept over shapes -- the same functions the HIP launches call."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_math_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.fail("g++ not found: the image is expected to have it")
    exe = tmp_path / "launch_math_test"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-Wall", "-Werror", os.path.join(REPO, "tests", "host", "launch_math_test.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "launch math: 0 failures" in run.stdout
