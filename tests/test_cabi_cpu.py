"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/azhip.h declares (no compute is launched without a GPU)."""
import ctypes

import pytest
import torch

from activezero_amd import _lib, build


@pytest.fixture(scope="module")
def handle():
    build.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_typed(handle):
    declared = _lib.declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in azhip.h but not exported"
    assert set(declared) == set(_lib._SIGS), "ctypes signatures out of sync with azhip.h"


def test_error_strings_and_version(handle):
    assert handle.az_abi_version() == _lib.expected_abi_version()
    assert handle.az_strerror(0) == b"AZ_OK"
    assert handle.az_strerror(-4) == b"AZ_EUNSUPPORTED"


def test_a_library_of_another_abi_version_is_refused(handle, monkeypatch):
    """a stale libazhip.so (or AZ_LIB_PATH variant) would take shifted arguments: lib() must raise, not load it"""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "expected_abi_version", lambda: handle.az_abi_version() + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib() is not None


def test_argument_validation_happens_before_any_launch(handle):
    # null pointers / bad dims are rejected on the host: safe without a GPU
    assert handle.az_warp_scatter(None, None, None, 1, 1, 4, 4, 1, None) == -2
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert handle.az_warp_scatter(p, p, p, 1, 1, 0, 4, 1, None) == -1
    assert handle.az_cost_volume_fwd(p, p, p, 1, 32, 0, 4, 4, None) == -1
    assert handle.az_softargmin_fwd(p, None, None, 1, 4, 4, 4, None) == -2
    assert handle.az_patch_reproj_fwd(p, p, p, p, None, 1, 1, 8, 8, 4, -1.0, None) == -1
    assert handle.az_patch_reproj_fwd(p, p, p, p, None, 1, 1, 8, 8, 17, -1.0, None) == -4


def test_ops_refuse_cpu_tensors():
    from activezero_amd import ops

    with pytest.raises(RuntimeError, match="GPU"):
        ops.softargmin(torch.zeros(1, 1, 4, 4, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.cost_volume(torch.zeros(1, 32, 4, 8), torch.zeros(1, 32, 4, 8), 2)


def test_dropin_module_surface():
    from activezero_amd.nets.psmnet import psmnet, psmnet_3, psmnet_submodule_3
    from activezero_amd.utils import reprojection, warp_ops

    for name in ("hourglass", "PSMNet"):
        assert hasattr(psmnet, name) and hasattr(psmnet_3, name)
    for name in ("convbn", "conv", "convbn_3d", "BasicBlock", "DisparityRegression",
                 "FeatureExtraction"):
        assert hasattr(psmnet_submodule_3, name)
    for name in ("apply_disparity", "get_reprojection_error", "get_reprojection_error_old",
                 "get_reproj_error_patch", "get_reprojection_error_diff_ratio",
                 "local_contrast_norm"):
        assert hasattr(reprojection, name)
    assert hasattr(warp_ops, "apply_disparity_cu")
