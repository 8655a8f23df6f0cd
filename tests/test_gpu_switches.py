"""Every route that only an A/B switch reaches (DESIGN.md, table "A/B switches") is run through existing parity tests in a
child process with the switch set: a route that ships is a route that is tested.  The switches are read once per
process (az_options.h / module import), hence the child processes -- one at a time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # (environment, test file, -k expression)
    ({"AZ_CONV_ROLL": "0"}, "tests/test_gpu_conv3d.py", "convbn3d_golden and bf16x6"),                # V0 layers on az_conv3d_m128.hip
    ({"AZ_WGRAD_R16": "0"}, "tests/test_gpu_conv3d.py", "residual_relu_train and bf16x6"),             # one-kd-per-wave stride-1 weight gradient
    ({"AZ_WGRAD_R16": "0", "AZ_WGRAD_FW": "0"}, "tests/test_gpu_conv3d.py", "residual_relu_train and bf16x6"),
    ({"AZ_WGRAD_R16": "1"}, "tests/test_gpu_conv3d.py", "convbn3d_golden and bf16x6"),
    ({"AZ_BN_BWD_FUSED": "0"}, "tests/test_gpu_conv3d.py", "convbn3d_golden and f16x3"),               # three-launch BatchNorm backward
    ({"AZ_BWD_F16": "0"}, "tests/test_gpu_conv3d.py", "convbn3d_golden and bf16x6"),                   # bf16x6 gradients
    ({"AZ_CONV_MAP": "0"}, "tests/test_gpu_conv3d.py", "conv_stride2_vs_torch"),                       # linear block -> tile map
    ({"AZ_CONV2D_ROLL": "0"}, "tests/test_gpu_conv2d.py", "same"),                                     # 32/64-channel 3x3 layers on K13
    ({"AZ_CONV2D_ROLL_NT4": "0"}, "tests/test_gpu_conv2d_roll.py", ""),
    ({"AZ_CONV2D_WGRAD_R16": "0"}, "tests/test_gpu_conv2d.py", "same"),
    ({"AZ_CORR_FP32": "1"}, "tests/test_gpu_raft_corr.py", ""),
    ({"AZ_PATCH_TILED": "0"}, "tests/test_gpu_kernels.py", "patch"),
    ({"AZ_PATCH_K": "1"}, "tests/test_gpu_kernels.py", "patch"),
    ({"AZ_WGRAD_R16_WGS": "64", "AZ_ROLL_SEGLEN": "5"}, "tests/test_gpu_conv3d.py", "convbn3d_golden"),
    ({"AZ_WGRAD_S2R16": "0"}, "tests/test_gpu_conv3d.py", "weight_grad_stride2 and f16x3"),            # stride-2 f16x3 weight gradient, one kd per wave
    ({"AZ_WGRAD_R16_XCD": "0"}, "tests/test_gpu_conv3d.py", "residual_relu_train and f16x3"),           # K4w columns in linear order
    ({"AZ_CONV_T2ROLL": "0"}, "tests/test_gpu_conv3d.py", "(deconv or hourglass_golden) and f16x3"),       # transposed 64 -> 32 on az_conv3d_t2.hip
    ({"AZ_CONV_S2ROLL": "0"}, "tests/test_gpu_conv3d.py", "(conv_stride2 or hourglass_golden or presplit_routing) and f16x3 or presplit_routing"),  # stride-2 32 -> 64 on the gather kernel
    ({"AZ_S2ROLL_SEGLEN": "1"}, "tests/test_gpu_s2roll.py", "forward or partials or input_gradient"),
    ({"AZ_CONV_ROLL64": "0"}, "tests/test_gpu_conv3d.py", "(conv_stride1 or convbn3d_golden or hourglass_golden) and f16x3"),        # stride-1 64 -> 64 on the gather kernel                               # one output plane per depth segment
    ({"AZ_GRU_ASSEMBLE": "0"}, "tests/test_gpu_raft_gru.py", "golden or fused_training_node"),      # the GRU's input rows by torch slice assignments
    ({"AZ_LOOKUP_ACC": "0"}, "tests/test_gpu_raft_corr.py", "golden or vs_torch_ops"),                # one gradient buffer per lookup and level
    ({"AZ_CONV2D_ROLL_H": "0"}, "tests/test_gpu_conv2d_roll.py", ""),                                      # f16x3 64-channel 2-D layers on conv2d_roll_kernel<.., 4, 1>
    # the Python-level switches of the two-stream backward (overlap.py, conv2d.py)
    ({"AZ_SIDE_RELEASE": "record"}, "tests/test_gpu_overlap.py", "in_order_pass or partial_backward or two_forward"),  # operands released through record_stream
    ({"AZ_SIDE_PRIORITY": "-1"}, "tests/test_gpu_overlap.py", "in_order_pass"),             # side stream at the other HIP priority
    ({"AZ_2D_WGRAD_OVERLAP": "0"}, "tests/test_gpu_overlap.py", "in_order_pass"),           # 2-D weight gradients in order
    ({"AZ_PRESPLIT": "0"}, "tests/test_gpu_conv3d.py", "(convbn3d_golden or residual_relu_train or hourglass_golden) and f16x3"),  # fp32 gradient operands, split by the consumers
    ({"AZ_PACK_PLAN": "0"}, "tests/test_gpu_psmnet.py", "train and not d192 and not full_size"),           # every weight image packed by its own launch
    ({"AZ_WGRAD_R16_WIDE": "1"}, "tests/test_gpu_conv3d.py", "(residual_relu_train or convbn3d_golden or hourglass_golden or presplit) and f16x3"),  # taps split over the waves, 32x32x16 tiles
    ({"AZ_CONV2D_WGRAD_W64": "0"}, "tests/test_gpu_conv2d.py", "same"),                                    # 64 -> 64 2-D weight gradients as 2 x 2 tiles of 32 x 32
    ({"AZ_GRAD_HANDOVER": "0"}, "tests/test_gpu_conv3d.py", "hourglass_golden or handover"),             # every consumer returns its own gradient, the engine adds
    ({"AZ_WGRAD_DEFER": "0"}, "tests/test_gpu_overlap.py", "in_order_pass or partial_backward"),            # per-layer workspace memset + unpack on the side stream
    ({"AZ_DEBUG_AMAX": "1"}, "tests/test_gpu_conv3d.py", "(convbn3d_golden or residual_relu_train) and f16x3"),  # every attached amax checked against a fresh pass
]


@pytest.mark.parametrize("env,path,expr", CASES, ids=[" ".join(f"{k}={v}" for k, v in c[0].items()) for c in CASES])
def test_route_behind_switch(env, path, expr):
    cmd = [sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    if expr:
        cmd += ["-k", expr]
    r = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


def test_switches_are_read_once_into_the_options_struct():
    """az_option reports what the library holds; changing the environment afterwards changes nothing"""
    code = ("import os; os.environ['AZ_WGRAD_R16_WGS']='77'; from activezero_amd import _lib; L=_lib.lib(); "
            "a=L.az_option(b'AZ_WGRAD_R16_WGS'); os.environ['AZ_WGRAD_R16_WGS']='5'; b=L.az_option(b'AZ_WGRAD_R16_WGS'); "
            "print(a, b, L.az_option(b'AZ_PATCH_K'), L.az_option(b'NOPE'))")
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    a, b, k, nope = r.stdout.split()
    assert (a, b, k) == ("77", "77", "4") and int(nope) < 0
