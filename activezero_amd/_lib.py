"""ctypes loader of libazhip.so -- the only way the Python host code reaches the
HIP kernels.  There is NO fallback: if the library is missing or a call fails,
a RuntimeError is raised (the product path never routes through oracle/ or a
CPU restatement)."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZ_LIB_PATH") or os.path.join(HERE, "lib", "libazhip.so")  # AZ_LIB_PATH: kernel A/B experiments
HEADER = os.path.join(HERE, "..", "include", "azhip.h")

_lib = None

_C = ctypes
_PTR, _INT, _SIZE, _LL = _C.c_void_p, _C.c_int, _C.c_size_t, _C.c_longlong

# name -> argtypes (return type is int unless listed in _RESTYPE)
_SIGS = {
    "az_abi_version": [],
    "az_hbm_copy_probe": [_PTR, _PTR, _LL, _PTR],
    "az_option": [_C.c_char_p],
    "az_strerror": [_INT],
    "az_warp_scatter": [_PTR, _PTR, _PTR, _INT, _INT, _INT, _INT, _INT, _PTR],
    "az_cost_volume_fwd": [_PTR] * 3 + [_INT] * 5 + [_PTR],
    "az_cost_volume_bwd": [_PTR] * 3 + [_INT] * 5 + [_PTR],
    "az_cost_volume_fwd_ndhwc": [_PTR] * 3 + [_INT] * 5 + [_PTR],
    "az_cost_volume_bwd_ndhwc": [_PTR] * 3 + [_INT] * 5 + [_PTR],
    "az_softargmin_fwd": [_PTR] * 3 + [_INT] * 4 + [_PTR],
    "az_softargmin_bwd": [_PTR] * 5 + [_INT] * 4 + [_PTR],
    "az_warp_gather_fwd": [_PTR] * 3 + [_INT] * 4 + [_PTR],
    "az_warp_gather_bwd": [_PTR] * 5 + [_INT] * 4 + [_PTR],
    "az_patch_reproj_fwd": [_PTR] * 5 + [_INT] * 5 + [_C.c_float, _PTR],
    "az_patch_reproj_bwd": [_PTR] * 7 + [_INT] * 5 + [_C.c_float, _PTR],
    "az_patch_reproj_vis": [_PTR] * 3 + [_INT] * 5 + [_C.c_float, _PTR],
    "az_lcn": [_PTR] * 3 + [_INT] * 4 + [_C.c_float, _C.c_longlong, _PTR],
    "az_sum4": [_PTR] * 5 + [_LL, _PTR],
    "az_spp_upsample_fwd": [_PTR, _PTR] + [_INT] * 7 + [_PTR],
    "az_spp_upsample_bwd_workspace": [_INT] * 4,
    "az_spp_upsample_bwd": [_PTR, _PTR, _LL, _PTR] + [_INT] * 7 + [_PTR],
    "az_costconv_edge_width": [_INT, _INT],
    "az_costconv_num_classes": [_INT],
    "az_costconv_merge_fwd": [_PTR] * 5 + [_INT, _PTR],
    "az_costconv_merge_bwd": [_PTR] * 5 + [_INT, _PTR],
    "az_costconv_assemble_fwd": [_PTR] * 4 + [_INT] * 4 + [_PTR],
    "az_costconv_assemble_bwd": [_PTR] * 4 + [_INT] * 4 + [_PTR],
    "az_bn3d_stats_tiles": [_LL, _INT],
    "az_bn3d_stats": [_PTR] * 3 + [_LL, _INT, _PTR],
    "az_bn2d_workspace": [_INT, _LL, _INT],
    "az_bn2d_fwd": [_PTR] * 12 + [_LL, _INT, _INT, _LL, _INT, _C.c_float, _C.c_float, _PTR, _PTR, _PTR, _LL, _PTR, _PTR],
    "az_conv2d_stats_tiles": [_INT] * 4,
    "az_conv2d_fwd_stats": [_PTR] * 5 + [_INT] * 11 + [_PTR],
    "az_conv2d_roll_packed_floats": [_INT] * 2,
    "az_conv2d_roll_pack": [_PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _PTR],
    "az_conv2d_roll_fwd": [_PTR] * 6 + [_INT] * 6 + [_PTR],
    "az_conv2d_roll_stats_rows": [_INT] * 6,
    "az_conv2d_roll_fwd_stats": [_PTR] * 5 + [_INT] * 6 + [_PTR],
    "az_bn2d_bwd": [_PTR] * 5 + [_LL] + [_PTR] * 8 + [_INT, _INT, _LL, _INT, _PTR, _PTR],
    "az_conv2d_pack_weights_f16": [_PTR, _PTR, _PTR] + [_INT] * 4 + [_LL, _LL] + [_INT] * 3 + [_PTR],
    "az_conv2d_fwd_f16": [_PTR] * 8 + [_INT] * 12 + [_PTR],
    "az_conv2d_fwd_stats_f16": [_PTR] * 7 + [_INT] * 11 + [_PTR],
    "az_conv2d_pack_weights_bf16_flipped": [_PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _INT, _PTR],
    "az_conv2d_wgrad_bf16": [_PTR, _PTR, _LL, _PTR, _PTR] + [_INT] * 9 + [_PTR],
    "az_conv2d_wgrad_f16": [_PTR, _PTR, _LL] + [_PTR] * 4 + [_INT] * 12 + [_PTR],
    "az_conv2d_roll_pack_f16": [_PTR, _PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _PTR],
    "az_conv2d_roll_fwd_f16": [_PTR] * 8 + [_INT] * 6 + [_PTR],
    "az_conv2d_roll_fwd_stats_f16": [_PTR] * 7 + [_INT] * 6 + [_PTR],
    "az_disp_loss_fwd": [_PTR] * 6 + [_C.c_float, _C.c_float, _C.c_longlong, _PTR],
    "az_disp_loss_bwd": [_PTR] * 8 + [_C.c_float, _C.c_float, _PTR, _PTR] + [_C.c_float] * 3 + [_C.c_longlong, _PTR],
    "az_disp_metrics": [_PTR] * 7 + [_INT, _C.c_longlong, _PTR],
    "az_conv3d_packed_floats": [_INT, _INT, _INT],
    "az_conv3d_pack_weights": [_PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _INT, _PTR],
    "az_conv3d_num_tiles": [_INT] * 5,
    "az_conv3d_stats_tiles": [_INT] * 8,
    "az_conv3d_fwd": [_PTR] * 7 + [_INT] * 10 + [_PTR],
    "az_conv3d_fwd_stats": [_PTR] * 6 + [_INT] * 9 + [_PTR],
    "az_conv3d_wgrad_workspace": [_INT, _INT],
    "az_conv3d_wgrad": [_PTR, _PTR, _LL, _PTR, _PTR] + [_INT] * 11 + [_PTR],
    "az_absmax": [_PTR, _PTR, _LL, _PTR],
    "az_pack_f16_multi": [_PTR, _PTR, _PTR, _INT, _INT, _PTR],
    "az_wgrad_unpack_multi": [_PTR, _PTR, _PTR, _INT, _INT, _PTR],
    "az_conv3d_packed_floats_f16": [_INT, _INT],
    "az_conv3d_f16_layout": [_INT, _INT, _INT],
    "az_conv3d_pack_weights_f16": [_PTR, _PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _INT, _PTR],
    "az_conv3d_fwd_f16": [_PTR] * 5 + [_INT] + [_PTR] * 3 + [_INT] * 8 + [_PTR],
    "az_conv3d_fwd_f16_split_ok": [_INT] * 7,
    "az_conv3d_wgrad_f16_split_ok": [_INT] * 10,
    "az_conv3d_stats_tiles_f16": [_INT] * 7,
    "az_conv3d_fwd_stats_f16": [_PTR] * 7 + [_INT] * 7 + [_PTR],
    "az_conv3d_wgrad_f16": [_PTR, _PTR, _LL] + [_PTR] * 4 + [_INT] * 11 + [_PTR],
    "az_conv3d_c1_fwd": [_PTR] * 6 + [_INT] * 4 + [_PTR],
    "az_conv3d_c1_dgrad": [_PTR] * 3 + [_INT] * 4 + [_PTR],
    "az_conv3d_c1_wgrad": [_PTR] * 5 + [_INT] * 4 + [_PTR],
    "az_bn3d_finalize": [_PTR] * 10 + [_LL, _INT, _C.c_float, _C.c_float, _PTR, _PTR, _LL, _PTR],
    "az_bn3d_finalize_scratch": [_INT],
    "az_bn3d_eval_affine": [_PTR] * 6 + [_C.c_float, _INT, _PTR],
    "az_bn3d_apply": [_PTR] * 5 + [_INT, _LL, _INT, _PTR, _PTR],
    "az_bn3d_bwd_workspace": [_LL, _INT],
    "az_bn3d_bwd": [_PTR] * 6 + [_LL] + [_PTR] * 8 + [_INT, _LL, _INT, _PTR, _INT, _PTR],
    "az_add_relu": [_PTR] * 3 + [_INT, _LL, _PTR, _PTR],
    "az_conv2d_packed_floats": [_INT] * 4,
    "az_conv2d_pack_weights": [_PTR, _PTR] + [_INT] * 4 + [_LL, _LL] + [_INT] * 3 + [_PTR],
    "az_conv2d_fwd": [_PTR] * 6 + [_INT] * 12 + [_PTR],
    "az_rows_concat": [_PTR, _LL, _LL, _INT, _PTR, _PTR, _PTR, _PTR],
    "az_rows_slice_to_image": [_PTR, _PTR, _LL, _LL, _INT, _INT, _INT, _PTR],
    "az_gru_rh": [_PTR] * 3 + [_LL, _INT, _INT, _PTR],
    "az_gru_out": [_PTR] * 4 + [_LL, _INT, _INT, _PTR],
    "az_gru_bwd1": [_PTR] * 7 + [_LL, _INT, _INT, _PTR, _PTR, _PTR],
    "az_gru_bwd2": [_PTR] * 5 + [_LL, _INT, _INT, _PTR, _PTR],
    "az_conv2d_pack_weights_h1": [_PTR, _PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _PTR],
    "az_conv2d_h1_fwd": [_PTR] * 9 + [_INT] * 11 + [_PTR],
    "az_conv2d_wgrad_h1": [_PTR, _PTR, _LL] + [_PTR] * 4 + [_INT] * 9 + [_PTR],
    "az_gru_bwd3": [_PTR] * 5 + [_LL, _INT, _INT, _PTR],
    "az_conv2d_pack_weights_bf16": [_PTR, _PTR, _INT, _INT, _LL, _LL, _INT, _INT, _PTR],
    "az_conv2d_bf16_fwd": [_PTR] * 7 + [_INT] * 11 + [_PTR],
    "az_conv2d_wgrad_workspace": [_INT] * 4,
    "az_conv2d_wgrad": [_PTR, _PTR, _LL, _PTR, _PTR] + [_INT] * 12 + [_PTR],
    "az_im2col_s2k3": [_PTR, _PTR] + [_INT] * 5 + [_PTR],
    "az_col2im_s2k3": [_PTR, _PTR] + [_INT] * 5 + [_PTR],
    "az_ir_pattern_workspace": [_INT] * 4,
    "az_ir_pattern": [_PTR, _PTR, _LL, _PTR, _PTR] + [_INT] * 4 + [_C.c_float, _PTR],
    "az_corr1d_volume": [_PTR] * 3 + [_INT] * 5 + [_PTR],
    "az_corr1d_volume_bwd": [_PTR] * 5 + [_INT] * 5 + [_PTR],
    "az_corr1d_pool": [_PTR, _PTR, _LL, _INT, _PTR],
    "az_corr1d_pool_bwd": [_PTR, _PTR, _LL, _INT, _PTR],
    "az_corr1d_lookup_fwd": [_PTR] * 3 + [_INT] * 8 + [_PTR],
    "az_corr1d_lookup_bwd": [_PTR] * 3 + [_INT] * 8 + [_PTR],
    "az_corr1d_lookup_bwd_acc": [_PTR] * 3 + [_INT] * 8 + [_PTR],
}
_RESTYPE = {"az_strerror": _C.c_char_p, "az_conv3d_num_tiles": _LL, "az_conv3d_stats_tiles": _LL, "az_conv2d_roll_packed_floats": _LL, "az_conv2d_roll_stats_rows": _LL, "az_conv3d_packed_floats": _LL, "az_conv3d_packed_floats_f16": _LL, "az_conv3d_stats_tiles_f16": _LL,
            "az_conv3d_wgrad_workspace": _LL, "az_bn3d_bwd_workspace": _LL, "az_spp_upsample_bwd_workspace": _LL,
            "az_bn3d_stats_tiles": _LL, "az_bn2d_workspace": _LL,
            "az_conv2d_packed_floats": _LL, "az_conv2d_wgrad_workspace": _LL, "az_ir_pattern_workspace": _LL,
            "az_bn3d_finalize_scratch": _LL, "az_conv2d_stats_tiles": _LL}


def declared_symbols():
    """Every function name include/azhip.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", text)))


def expected_abi_version():
    """AZ_ABI_VERSION of include/azhip.h: the signatures _SIGS was written against"""
    m = re.search(r"#define\s+AZ_ABI_VERSION\s+(\d+)", open(HEADER).read())
    if m is None:
        raise RuntimeError(f"{HEADER} defines no AZ_ABI_VERSION")
    return int(m.group(1))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m activezero_amd.build` "
                "(there is no non-HIP fallback)")
        handle = _C.CDLL(LIB_PATH)
        # a stale build (or a stale variant picked up through AZ_LIB_PATH) would read shifted arguments: refuse it
        handle.az_abi_version.argtypes, handle.az_abi_version.restype = [], _INT
        have, want = handle.az_abi_version(), expected_abi_version()
        if have != want:
            raise RuntimeError(f"{LIB_PATH} has ABI version {have}, include/azhip.h declares {want}: rebuild it "
                               "(`python -m activezero_amd.build --force`)")
        for name, args in _SIGS.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _INT)
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed: {lib().az_strerror(code).decode()} ({code})")
