"""ORACLE (test infrastructure, never the product path).

Scatter-warp oracle: `apply_disparity_cu` of /root/reference/utils/warp_ops.py:55-95.
Two restatements that must agree bit for bit:
  * warp_scatter_oracle.c (compiled by oracle/Makefile -> oracle/build/libaz_oracle.so)
  * a pure-numpy row loop (small cases / when the .so has not been built).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libaz_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None and os.path.exists(_SO):
        _lib = ctypes.CDLL(_SO)
        _lib.az_oracle_warp_scatter.restype = ctypes.c_int
        _lib.az_oracle_warp_scatter.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4
    return _lib


def warp_scatter_numpy(src, disp):
    """src [N,C,H,W] float32, disp [N,H,W] int32 -> [N,C,H,W] (serial semantics)."""
    n, c, h, w = src.shape
    assert (disp >= 0).all() or (disp <= 0).all()
    pos = bool((disp >= 0).all())
    out = np.zeros_like(src)
    order = range(w - 1, -1, -1) if pos else range(w)
    for b in range(n):
        for y in range(h):
            drow = disp[b, y]
            for j in order:
                t = j + int(drow[j])
                if (pos and t < w) or (not pos and t > -1):
                    out[b, :, y, t] = src[b, :, y, j]
    return out


def warp_scatter_c(src, disp):
    lib = _load()
    if lib is None:
        raise RuntimeError("oracle/build/libaz_oracle.so missing: run `make -C oracle`")
    src = np.ascontiguousarray(src, dtype=np.float32)
    disp = np.ascontiguousarray(disp, dtype=np.int32)
    n, c, h, w = src.shape
    out = np.empty_like(src)
    rc = lib.az_oracle_warp_scatter(out.ctypes.data, src.ctypes.data, disp.ctypes.data, n, c, h, w)
    if rc != 0:
        raise AssertionError("disp must be all >= 0 or all <= 0 (warp_ops.py:73-77)")
    return out


def apply_disparity_cu_oracle(img, disp):
    """torch-facing oracle of apply_disparity_cu(img [N,C,H,W] f32, disp [N,(1,)H,W] i32)."""
    assert disp.dtype == torch.int32
    d = disp.detach().cpu().numpy().reshape(img.shape[0], img.shape[2], img.shape[3])
    s = img.detach().cpu().contiguous().numpy().astype(np.float32)
    fn = warp_scatter_c if _load() is not None else warp_scatter_numpy
    return torch.from_numpy(fn(s, d)).to(img.device)
