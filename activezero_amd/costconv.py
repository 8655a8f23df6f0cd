"""dres0[0] on the concat cost volume without the cost volume (reference nets/psmnet/psmnet_3.py:
149-166): the volume is constant along d in its left-feature half and an x-shift of the right feature map
in the other half, so its 3x3x3 convolution factors into 2-D convolutions of the two feature maps
(csrc/az_costconv.hip has the derivation) -- 80 GFLOP at B=4 instead of 693, no 1.6 GB volume, no
1.6 GB volume gradient.

  merged kernels (differentiable tensor ops on the Conv3d weight)
    K_L[c, dl][o, i, kh, kw] = sum over kd in class c with kd - kw <= delta(dl) of W[o, i, kd, kh, kw]
    K_R[c, xb][o, i, kh, j]  = sum over kd in class c, kw with kw - kd = j - 2 (and kw <= 1 if xb) of W[o, 32+i, kd, kh, kw]
  F = conv2d(L, K_L, pad 1), G = conv2d(pad_left2(R), K_R, pad (1,2))        (conv2d.py: az_conv2d_fwd / _wgrad)
  raw = assemble(F, G)                                                       (az_costconv_assemble_*)
"""
import torch
import torch.nn.functional as F_

from . import conv2d, profiler
from .ops import _call, _chk, _p, _stream

NDL = 5
_MASKS = {}


def num_classes(ndisp):
    """depth classes present (csrc/az_costconv.hip cc_class): first / middle / last plane."""
    return min(int(ndisp), 3)


def _masks(device, ndisp):
    """0/1 selection tensors: ML[c, dl, kd, kw], MR[c, xb, kd, kw, j]."""
    ncls = num_classes(ndisp)
    key = (str(device), ncls)
    if key not in _MASKS:
        ml = torch.zeros(ncls, NDL, 3, 3)
        mr = torch.zeros(ncls, 2, 3, 3, 5)
        # which depth taps exist per class: D >= 3: first {1,2}, middle {0,1,2}, last {0,1}; D = 2: first, last; D = 1: {1}
        taps = {1: [(1,)], 2: [(1, 2), (0, 1)], 3: [(1, 2), (0, 1, 2), (0, 1)]}[ncls]
        for c in range(ncls):
            has = {kd: (kd in taps[c]) for kd in range(3)}
            for kd in range(3):
                if not has[kd]:
                    continue
                for kw in range(3):
                    for dl in range(NDL):
                        delta = dl - 2  # dl = 4 stands for every delta >= 2: all taps pass
                        if kd - kw <= delta:
                            ml[c, dl, kd, kw] = 1.0
                    for xb in range(2):
                        if xb == 1 and kw == 2:
                            continue  # x = W-1: the tap reads x' = W, outside the volume
                        mr[c, xb, kd, kw, kw - kd + 2] = 1.0
        _MASKS[key] = (ml.to(device), mr.to(device))
    return _MASKS[key]


_MERGED_CACHE = {}


def _merged_kernels(weight, ndisp):
    """(K_L bulk [ncls*32,32,3,3], K_L edge [ncls*128,32,3,3], K_R [ncls*64,32,3,5]) of a [32,64,3,3,3] weight;
    differentiable.  Memoised between no_grad forwards on the weight's version counter."""
    from .conv3d import _cache_get, _cache_put
    ncls = num_classes(ndisp)
    key = (weight.data_ptr(), weight._version, weight.device.index, ncls) if not torch.is_grad_enabled() else None
    if key is not None:
        hit = _cache_get(_MERGED_CACHE, key)
        if hit is not None:
            return hit[:3]
    cl = torch.channels_last
    ml, mr = _masks(weight.device, ndisp)
    # kl = einsum("oidhw,cedw->ceoihw", weight[:, :32], ml), kr = einsum("oidhw,cedwj->ceoihj", weight[:, 32:], mr)
    if weight.is_cuda:
        kl, kr = _Merge.apply(weight, ml, mr, ncls)                      # [cls, dl, o, i, 3, 3], [cls, xb, o, i, 3, 5]
    else:  # (tests/test_costconv_cpu.py checks this algebra on the host; every kernel downstream rejects CPU tensors)
        kl = torch.einsum("oidhw,cedw->ceoihw", weight[:, :32], ml)
        kr = torch.einsum("oidhw,cedwj->ceoihj", weight[:, 32:], mr)
    out = (kl[:, 4].reshape(ncls * 32, 32, 3, 3).contiguous(memory_format=cl),
           kl[:, :4].reshape(ncls * 4 * 32, 32, 3, 3).contiguous(memory_format=cl),
           kr.reshape(ncls * 2 * 32, 32, 3, 5).contiguous(memory_format=cl))
    if key is not None:
        _cache_put(_MERGED_CACHE, key, out + (weight,), 16)  # (keeps the source alive: its address stays unique)
    return out


class _Merge(torch.autograd.Function):
    """the two masked depth sums of the Conv3d weight (module docstring) on az_costconv_merge_fwd / _bwd -- weight-space
    einsums of 55 K elements that were the step's last rocBLAS launches"""

    @staticmethod
    def forward(ctx, weight, ml, mr, ncls):
        w = _chk(weight.detach().contiguous(), "weight")
        kl = w.new_empty(ncls, NDL, 32, 32, 3, 3)
        kr = w.new_empty(ncls, 2, 32, 32, 3, 5)
        with torch.cuda.device(w.device):
            _call("az_costconv_merge_fwd", _p(kl), _p(kr), _p(w), _p(ml), _p(mr), ncls, _stream())
        ctx.save_for_backward(ml, mr)
        ctx.ncls = ncls
        return kl, kr

    @staticmethod
    def backward(ctx, gkl, gkr):
        ml, mr = ctx.saved_tensors
        gkl, gkr = _chk(gkl.contiguous(), "grad K_L"), _chk(gkr.contiguous(), "grad K_R")
        gw = gkl.new_empty(32, 64, 3, 3, 3)
        with torch.cuda.device(gkl.device):
            _call("az_costconv_merge_bwd", _p(gw), _p(gkl), _p(gkr), _p(ml), _p(mr), ctx.ncls, _stream())
        return gw, None, None, None


class _Assemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fb, fe, g, ndisp):
        fb, fe, g = _chk(fb, "F_bulk"), _chk(fe, "F_edge"), _chk(g, "G")  # channels-last rows
        b, h, w, _ = fb.shape
        out = fb.new_empty(b, ndisp, h, w, 32)
        with torch.cuda.device(fb.device):
            with profiler.scope("costconv_assemble", bytes=4.0 * out.numel(), bound="hbm"):
                _call("az_costconv_assemble_fwd", _p(out), _p(fb), _p(fe), _p(g), b, ndisp, h, w, _stream())
        ctx.dims = (b, ndisp, h, w, fe.shape[2], fb.shape[3] // 32)
        return out

    @staticmethod
    def backward(ctx, gy):
        b, d, h, w, xe, ncls = ctx.dims
        gy = _chk(gy.contiguous(), "grad_out")
        dfb = gy.new_empty(b, h, w, ncls * 32)
        dfe = gy.new_empty(b, h, xe, ncls * 4 * 32)
        dg = gy.new_empty(b, h, w + 2, ncls * 2 * 32)
        with torch.cuda.device(gy.device):
            with profiler.scope("costconv_assemble_bwd", bytes=8.0 * gy.numel(), bound="hbm"):
                _call("az_costconv_assemble_bwd", _p(dfb), _p(dfe), _p(dg), _p(gy), b, d, h, w, _stream())
        return dfb, dfe, dg, None


def costvol_conv(feat_l, feat_r, ndisp, weight, arith=None):
    """conv3d(concat_cost_volume(feat_l, feat_r, ndisp), weight, padding=1) as [B,ndisp,h,w,32] (NDHWC).
    feat_*: [B,32,h,w] (channels_last preferred); weight: [32,64,3,3,3]."""
    if tuple(weight.shape) != (32, 64, 3, 3, 3) or feat_l.shape[1] != 32 or feat_l.shape != feat_r.shape:
        raise RuntimeError("costvol_conv expects two [B,32,h,w] feature maps and a [32,64,3,3,3] weight")
    from . import _lib
    w = feat_l.shape[-1]
    xe = _lib.lib().az_costconv_edge_width(int(ndisp), w)
    cl = torch.channels_last
    kl_bulk, kl_edge, kr = _merged_kernels(weight, ndisp)
    fl = feat_l.contiguous(memory_format=cl)
    fb = conv2d.conv_same(fl, kl_bulk)
    # the delta < 2 maps are read at x = d + delta <= ndisp only: convolve the first xe (+1 halo) columns
    fe = conv2d.conv_same(fl[..., :min(w, xe + 1)].contiguous(memory_format=cl), kl_edge)[..., :xe]
    rp = F_.pad(feat_r, (2, 0)).contiguous(memory_format=cl)
    g = conv2d.conv_same(rp, kr)
    rows = lambda t: t.permute(0, 2, 3, 1).contiguous()  # [B,C,h,w] -> [B,h,w,C]
    return _Assemble.apply(rows(fb), rows(fe), rows(g), int(ndisp))
