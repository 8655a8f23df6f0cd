"""Integer scatter warp on the GPU -- drop-in for the reference module
utils/warp_ops.py (`apply_disparity_cu`).

The reference JIT-compiles a CUDA-C string through NVRTC/cupy and walks every
image row with one thread (utils/warp_ops.py:20-52, 80-93).  Here the call goes
through the C ABI (az_warp_scatter, include/azhip.h) to a wavefront-per-row HIP
kernel with the same collision rule; same signature, same preconditions, same
result tensor (new, zero-filled where nothing lands).
"""
import torch

from activezero_amd import ops


def apply_disparity_cu(img: torch.Tensor, disp: torch.Tensor, sign=None):
    """
    :param img: tensor to warp, (N, C, H, W) float32, contiguous, on the GPU
    :param disp: (N, H, W) or (N, 1, H, W) int32, all >= 0 or all <= 0
    :param sign: optional +1 / -1 to skip the device-synchronising sign check
                 (extension; the reference always checks, warp_ops.py:73-77)
    """
    assert img.is_contiguous() and disp.is_contiguous()
    assert img.device.type == disp.device.type == "cuda"
    assert disp.dtype == torch.int
    if sign is None:
        if torch.all(disp >= 0):
            sign = 1
        else:
            assert torch.all(disp <= 0)
            sign = -1
    return ops.warp_scatter(img, disp, sign)
