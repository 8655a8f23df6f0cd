"""GPU twins of the reference's per-item pattern helpers (datasets/dataset_utils.py) -- SURVEY.md 8f-4.
The reference computes them with numpy / cv2 inside DataLoader workers, one image at a time; here a whole
batch is processed on the device by az_ir_pattern (csrc/az_ir_pattern.hip)."""
import torch

from activezero_amd import _lib
from activezero_amd.ops import _call, _chk, _p, _stream


def get_smoothed_ir_pattern2(img_ir, img, ks=11, threshold=0.005):
    """dataset_utils.py:33-46 for [H,W] or [B,H,W] float32 CUDA tensors -> binary pattern of the same shape."""
    squeeze = img_ir.dim() == 2
    a = _chk(img_ir.reshape(-1, *img_ir.shape[-2:]).contiguous(), "img_ir")
    b_ = _chk(img.reshape(-1, *img.shape[-2:]).contiguous(), "img")
    if a.shape != b_.shape:
        raise RuntimeError("img_ir and img must have identical shapes")
    n, h, w = a.shape
    ws_bytes = _lib.lib().az_ir_pattern_workspace(n, h, w, int(ks))
    if ws_bytes < 0:
        raise RuntimeError("get_smoothed_ir_pattern2: image smaller than the smoothing window")
    ws = a.new_empty((ws_bytes + 3) // 4)
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _call("az_ir_pattern", _p(out), _p(ws), ws_bytes, _p(a), _p(b_), n, h, w, int(ks), float(threshold), _stream())
    return out[0] if squeeze else out.view(img_ir.shape)
