"""CPU: the drop-in mechanics of INTEGRATION.md section 2, executed.

1. With this repository and its `activezero_amd/` directory IN FRONT of a reference-shaped tree on
   sys.path, `nets.psmnet.psmnet_3`, `nets.psmnet.psmnet`, `utils.warp_ops`, `utils.reprojection` resolve to
   the build while `nets.adapter`, `nets.dispnet`, `utils.losses` ... keep resolving from the reference
   tree (pkgutil.extend_path in activezero_amd/{nets,utils}/__init__.py).  The reference checkout itself is
   not available on the GPU box, so the test builds a throw-away tree with the reference's package layout
   (empty modules of the same names); when /root/reference exists (the build container) the real tree is
   checked as well, by FILE LOCATION only -- nothing of it is imported or executed.
2. The argument checks of the 2-D convolution C ABI run on the host (no GPU needed).
3. The oracle reproduces the reference's D = 192 golden (tools/make_goldens.py g11)."""
import ctypes
import importlib.util
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = textwrap.dedent("""
    import importlib.util, json, sys
    names = ["nets.psmnet.psmnet_3", "nets.psmnet.psmnet", "nets.psmnet.psmnet_submodule_3", "nets.psmnet.psmnet_submodule",
             "utils.warp_ops", "utils.reprojection", "utils.cascade_metrics", "nets.adapter", "nets.dispnet.dispnet",
             "utils.losses", "utils.reduce"]
    out = {}
    for n in names:
        spec = importlib.util.find_spec(n)      # locates the module file WITHOUT importing the leaf module
        out[n] = spec.origin if spec else None
    print(json.dumps(out))
""")


def _resolve(ref_root):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([REPO, os.path.join(REPO, "activezero_amd"), ref_root])
    r = subprocess.run([sys.executable, "-c", _PROBE], capture_output=True, text=True, env=env, check=True, cwd="/")
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def _check(found, ref_root):
    build = os.path.join(REPO, "activezero_amd") + os.sep
    for n in ("nets.psmnet.psmnet_3", "nets.psmnet.psmnet", "nets.psmnet.psmnet_submodule_3",
              "nets.psmnet.psmnet_submodule", "utils.warp_ops", "utils.reprojection", "utils.cascade_metrics"):
        assert found[n] and found[n].startswith(build), (n, found[n])
    for n in ("nets.adapter", "nets.dispnet.dispnet", "utils.losses", "utils.reduce"):
        assert found[n] and found[n].startswith(ref_root), (n, found[n])


def test_overlay_resolution_on_a_reference_shaped_tree(tmp_path):
    ref = tmp_path / "ActiveZero"
    for rel in ("nets/__init__.py", "nets/adapter.py", "nets/dispnet/__init__.py", "nets/dispnet/dispnet.py",
                "nets/psmnet/__init__.py", "nets/psmnet/psmnet_3.py", "nets/psmnet/psmnet.py",
                "nets/psmnet/psmnet_submodule_3.py", "nets/psmnet/psmnet_submodule.py",
                "utils/__init__.py", "utils/losses.py", "utils/reduce.py", "utils/warp_ops.py",
                "utils/reprojection.py", "utils/cascade_metrics.py"):
        f = ref / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text("# stand-in with the reference's module name\n")
    _check(_resolve(str(ref)), str(ref))


@pytest.mark.skipif(not os.path.isdir("/root/reference/nets"), reason="reference checkout only exists in the build container")
def test_overlay_resolution_on_the_reference_checkout():
    _check(_resolve("/root/reference"), "/root/reference")


def test_conv2d_abi_argument_validation_on_host():
    from activezero_amd import _lib, build
    build.build()
    h = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert h.az_conv2d_packed_floats(32, 64, 3, 3) == 9 * 32 * 64 * 3 // 2
    assert h.az_conv2d_packed_floats(24, 64, 3, 3) < 0          # cin % 16
    assert h.az_conv2d_packed_floats(32, 48, 3, 3) < 0          # cout % 32
    assert h.az_conv2d_fwd(None, p, p, None, None, None, 0, 1, 8, 8, 32, 32, 32, 32, 0, 3, 3, 1, None) == -2
    assert h.az_conv2d_fwd(p, p, p, None, None, None, 0, 1, 8, 8, 24, 32, 32, 32, 0, 3, 3, 1, None) == -4
    assert h.az_conv2d_fwd(p, p, p, None, None, None, 0, 1, 8, 8, 32, 32, 16, 32, 0, 3, 3, 1, None) == -1  # stride < cin
    assert h.az_conv2d_fwd(p, p, p, None, None, None, 0, 1, 8, 8, 32, 32, 32, 32, 0, 7, 7, 1, None) == -4  # geometry
    assert h.az_conv2d_wgrad_workspace(64, 32, 3, 3) == 9 * 64 * 32 * 4
    assert h.az_conv2d_wgrad(p, p, 16, p, p, 1, 8, 8, 64, 32, 64, 32, 64, 32, 3, 3, 1, None) == -5  # workspace
    assert h.az_im2col_s2k3(p, p, 1, 3, 8, 8, 16, None) == -1   # Kp < 9 C


def test_oracle_reproduces_the_d192_reference_golden(golden):
    """g11 (maxdisp 192, one 256x512 pair, eval): the oracle in fp32 against the reference's fp32 output and
    in fp64 against the reference's fp64 output."""
    from oracle import psmnet_oracle as po
    from tests._weights import load_bn_buffers, load_procedural, seeded

    g = golden("g11_psmnet3_d192")
    st = int(g["pred_stride"])
    il, ir = seeded((1, 3, 256, 512), 1101, -2.0, 2.0), seeded((1, 3, 256, 512), 1102, -2.0, 2.0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    m = load_bn_buffers(load_procedural(po.PSMNetOracle(192, 3), "g11."), g).eval()
    with torch.no_grad():
        p32 = m(il, ir)[..., ::st, ::st].numpy()
        p64 = m.double()(il.double(), ir.double())[..., ::st, ::st].numpy()
    # same op sequence as the reference; thread-count-dependent summation order inside the convolutions
    # leaves rounding-level differences, amplified like any fp32 perturbation (|ref32 - ref64| = 8.7e-4)
    assert np.abs(p32 - g["eval32"]).max() <= 1e-3
    assert np.abs(p64 - g["eval64"]).max() <= 1e-9
