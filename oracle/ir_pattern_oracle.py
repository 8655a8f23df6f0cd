"""ORACLE (test infrastructure, never the product path).

CPU restatement (numpy, float64 like the reference's arrays) of datasets/dataset_utils.py:33-46
get_smoothed_ir_pattern2.  PARITY UNPINNED: the function's arithmetic is cv2.resize(..., INTER_AREA)
(opencv-python 4.5.3.56 / 4.5.5.62, environment.yaml:70-71 / requirements.txt:68-69), a third-party
dependency that is absent from this image and from /root/reference, and the reference holds no fixture
for it.  INTER_AREA is restated from OpenCV's published algorithm (modules/imgproc/src/resize.cpp:
computeResizeAreaTab + resizeArea_ for shrinking; the `area_mode` branch of the linear resize for
enlarging); everything else (abs difference, min-max normalisation, threshold) is the reference's own
numpy code, dataset_utils.py:38-46.  Only tests/ may import this file.
"""
import math

import numpy as np


def _area_tab(ssize, dsize):
    """rows (dst, src, weight) of OpenCV's computeResizeAreaTab for scale = ssize / dsize >= 1"""
    scale = ssize / dsize
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_shrink(src, hs, ws):
    h, w = src.shape
    mx = np.zeros((w, ws))
    for d, s, a in _area_tab(w, ws):
        mx[s, d] += a
    my = np.zeros((hs, h))
    for d, s, a in _area_tab(h, hs):
        my[d, s] += a
    return my @ (src.astype(np.float64) @ mx)


def _up_index(dsize, ssize):
    scale, inv = ssize / dsize, dsize / ssize
    idx, frac = np.zeros(dsize, np.int64), np.zeros(dsize)
    for d in range(dsize):
        s = math.floor(d * scale)
        fx = np.float32((d + 1) - (s + 1) * inv)
        fx = np.float32(0.0) if fx <= 0 else fx - np.floor(fx)
        if s >= ssize - 1:
            fx, s = np.float32(0.0), ssize - 1
        idx[d], frac[d] = s, fx
    return idx, frac


def resize_area_enlarge(src, h, w):
    hs, ws = src.shape
    iy, fy = _up_index(h, hs)
    ix, fx = _up_index(w, ws)
    ix1, iy1 = np.minimum(ix + 1, ws - 1), np.minimum(iy + 1, hs - 1)
    rows = src[:, ix] * (1 - fx)[None, :] + src[:, ix1] * fx[None, :]
    return rows[iy] * (1 - fy)[:, None] + rows[iy1] * fy[:, None]


def get_smoothed_ir_pattern2(img_ir, img, ks=11, threshold=0.005, return_margin=False):
    h, w = img_ir.shape
    hs, ws = int(h // ks), int(w // ks)
    diff = np.abs(img_ir - img)
    diff = (diff - np.min(diff)) / (np.max(diff) - np.min(diff))
    avg = resize_area_enlarge(resize_area_shrink(diff, hs, ws), h, w)
    ir = np.zeros_like(diff)
    diff2 = diff - avg
    ir[diff2 > threshold] = 1
    return (ir, diff2) if return_margin else ir
