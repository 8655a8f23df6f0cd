"""The 32-bit-offset kernels (depth-rolling 3-D, batch-walking 2-D) are routed around when a slice reaches 4 GiB
(ADVICE r3): the routing predicates are pure host logic."""
import torch

from activezero_amd import conv2d, conv3d


class _Shape:  # a stand-in with a tensor's shape (no 4 GiB allocation in a test)
    def __init__(self, *s):
        self.shape = torch.Size(s)


def test_conv3d_layout_skips_the_rolling_kernel_at_4gib():
    small, big = _Shape(1, 48, 136, 240, 32), _Shape(1, 192, 544, 960, 32)  # 0.2 GB / 12.8 GB per batch element
    assert conv3d._fits32(small, 32, 32) and not conv3d._fits32(big, 32, 32)
    assert conv3d._layout(conv3d.BF16X6, conv3d.CONV_S1, 32, True) == conv3d.BF16X6_R16
    assert conv3d._layout(conv3d.BF16X6, conv3d.CONV_S1, 32, False) == conv3d.BF16X6
    # exactly at the limit: d*h*w*c*4 == 0xffffff00 does not fit
    assert not conv3d._fits32(_Shape(1, 1, 1, 0xffffff00 // 128, 32), 32, 32)
    assert conv3d._fits32(_Shape(1, 1, 1, 0xffffff00 // 128 - 1, 32), 32, 32)


def test_conv2d_roll_route_checks_the_batch_size():
    ok = _Shape(8, 272, 480, 32)
    too_big = _Shape(64, 1024, 2048, 32)  # 64 * 1024 * 2048 * 256 B = 32 GiB
    assert conv2d._roll_ok(ok, 32, 32, 3, 3, 1)
    assert not conv2d._roll_ok(too_big, 32, 32, 3, 3, 1)


def test_transposed_roll_route_checks_input_and_output():
    v1 = _Shape(1, 24, 68, 120, 64)             # 50 MB in, 200 MB out per batch element
    big_out = _Shape(1, 96, 272, 480, 64)       # 3.2 GB in, 12.8 GB out
    mid = _Shape(1, 48, 136, 240, 64)           # 0.4 GB in, 1.6 GB out: fits
    edge = _Shape(1, 64, 256, 256, 64)          # 1.07 GB in, 4.29 GB out: the OUTPUT does not fit
    assert conv3d._fits32_transposed(v1, 64, 32) and conv3d._fits32_transposed(mid, 64, 32)
    assert not conv3d._fits32_transposed(big_out, 64, 32) and not conv3d._fits32_transposed(edge, 64, 32)
    assert conv3d._fits32_transposed(big_out, 64, 64)  # other channel counts do not take that kernel
