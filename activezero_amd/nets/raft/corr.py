"""RAFT-Stereo 1-D correlation block on the MI355X kernels -- drop-in for the `reg`
implementation of the reference module nets/raft/corr.py (`CorrBlock1D`, the default
cfg.MODEL.CORR_IMPLEMENTATION, configs/config.py:16).

  CorrBlock1D.corr(fmap1, fmap2)  -> az_corr1d_volume      (fp32 MFMA batched GEMM)
  pyramid (avg_pool2d [1,2])      -> az_corr1d_pool
  __call__(coords)                -> az_corr1d_lookup_fwd  (all 4 levels into one tensor)

Same constructor arguments, attributes (`num_levels`, `radius`, `corr_pyramid` with the
reference's [B*H*W1, 1, 1, W2/2^i] shapes) and return layout [B, levels*(2r+1), H, W1].
Gradients flow to fmap1/fmap2 (the lookup coordinate is detached by RAFT before the
call, raft_stereo.py, and receives none here).

The reference module's other classes need CUDA extensions that are not part of the
reference tree (`corr_sampler`, `alt_cuda_corr`) or are the numerically
different `alt` variant; they are exported as explicit "not targeted" stubs so that
`from nets.raft.corr import ...` in raft_stereo.py keeps working.
"""
import os

import torch

from activezero_amd import _lib, profiler
from activezero_amd.ops import _call, _chk, _p, _stream


class _Volume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2):
        f1, f2 = _chk(f1.contiguous(), "fmap1"), _chk(f2.contiguous(), "fmap2")
        b, c, h, w1 = f1.shape
        w2 = f2.shape[3]
        if f2.shape[:3] != f1.shape[:3]:
            raise RuntimeError("fmap1 / fmap2 must agree in batch, channels and height")
        corr = f1.new_empty(b, h, w1, w2)
        with torch.cuda.device(f1.device), profiler.scope("corr1d_volume", flops=2.0 * b * h * w1 * w2 * c,
                                                          bytes=4.0 * b * h * (c * (w1 + w2) + w1 * w2), bound="hbm"):
            _call("az_corr1d_volume", _p(corr), _p(f1), _p(f2), b, c, h, w1, w2, _stream())
        ctx.save_for_backward(f1, f2)
        return corr

    @staticmethod
    def backward(ctx, g):
        f1, f2 = ctx.saved_tensors
        g = _chk(g.contiguous(), "grad_corr")
        b, c, h, w1 = f1.shape
        w2 = f2.shape[3]
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(g.device), profiler.scope("corr1d_volume_bwd", flops=4.0 * b * h * w1 * w2 * c,
                                                         bytes=4.0 * b * h * (2 * c * (w1 + w2) + w1 * w2), bound="hbm"):
            _call("az_corr1d_volume_bwd", _p(g1), _p(g2), _p(g), _p(f1), _p(f2), b, c, h, w1, w2, _stream())
        return g1, g2


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src):
        src = _chk(src.contiguous(), "corr")
        w = src.shape[-1]
        rows = src.numel() // w
        dst = src.new_empty(*src.shape[:-1], w // 2)
        with torch.cuda.device(src.device):
            _call("az_corr1d_pool", _p(dst), _p(src), rows, w, _stream())
        ctx.dims = (tuple(src.shape), rows, w)
        return dst

    @staticmethod
    def backward(ctx, g):
        shape, rows, w = ctx.dims
        g = _chk(g.contiguous(), "grad_pool")
        gs = g.new_empty(shape)
        with torch.cuda.device(g.device):
            _call("az_corr1d_pool_bwd", _p(gs), _p(g), rows, w, _stream())
        return gs


class _LevelSums:
    """Gradient accumulators of a pyramid's levels for ONE backward pass (round 5).  Every lookup of a step reads the same
    pyramid, so the gradient of a level is the sum over the lookups: each _Lookup.backward scatters into the pass's buffer
    (az_corr1d_lookup_bwd_acc: the kernel's atomics add to what is there) and only the FIRST one to run hands the buffer to
    the engine; the others return None.  The engine runs the node that produced a level only after every lookup node that
    reads it has run, so by then the buffer holds the whole sum -- with 22 lookups and 4 levels that replaces 88 cleared
    buffers and 84 tensor additions per step by 4 buffers.  Keyed on the engine's graph-task id: a second backward pass
    over a retained graph starts from fresh buffers."""

    def __init__(self, n):
        self.task, self.bufs = None, [None] * n

    def _release(self):
        self.task, self.bufs = None, [None] * len(self.bufs)

    def take(self, i, like, shape):
        """(buffer of level i, True if this call created it: the caller then returns it as the gradient)"""
        task = torch._C._current_graph_task_id()
        if task != self.task:
            self.task, self.bufs = task, [None] * len(self.bufs)
            # (the buffers belong to the engine once handed over: let go of them when this backward pass ends)
            torch.autograd.Variable._execution_engine.queue_callback(self._release)
        fresh = self.bufs[i] is None
        if fresh:
            self.bufs[i] = like.new_zeros(shape)
        return self.bufs[i], fresh


# AZ_LOOKUP_ACC=0 (read once): one cleared gradient buffer per lookup and level, summed by the engine (A/B runs)
LOOKUP_ACC = os.environ.get("AZ_LOOKUP_ACC", "1") != "0"


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, sums, *levels):
        ctx.sums = sums if LOOKUP_ACC else None
        coords = _chk(coords.detach().contiguous(), "coords")
        b, two, h, w1 = coords.shape
        if two != 2:
            raise RuntimeError("coords must be [B,2,H,W]")
        taps = 2 * radius + 1
        out = coords.new_empty(b, taps * len(levels), h, w1)
        with torch.cuda.device(coords.device):
            for i, lvl in enumerate(levels):
                lvl = _chk(lvl, f"pyramid[{i}]")
                _call("az_corr1d_lookup_fwd", _p(out), _p(lvl), _p(coords), b, h, w1, lvl.shape[-1], radius,
                      i, i * taps, taps * len(levels), _stream())
        ctx.save_for_backward(coords)
        ctx.meta = (radius, [tuple(l.shape) for l in levels])
        return out

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        radius, shapes = ctx.meta
        g = _chk(g.contiguous(), "grad_lookup")
        b, _, h, w1 = coords.shape
        taps = 2 * radius + 1
        grads = []
        with torch.cuda.device(g.device):
            for i, shp in enumerate(shapes):
                if not ctx.needs_input_grad[3 + i]:
                    grads.append(None)
                    continue
                if ctx.sums is not None:
                    gl, fresh = ctx.sums.take(i, g, shp)
                    _call("az_corr1d_lookup_bwd_acc", _p(gl), _p(g), _p(coords), b, h, w1, shp[-1], radius, i,
                          i * taps, taps * len(shapes), _stream())
                    grads.append(gl if fresh else None)
                    continue
                gl = g.new_empty(shp)
                _call("az_corr1d_lookup_bwd", _p(gl), _p(g), _p(coords), b, h, w1, shp[-1], radius, i,
                      i * taps, taps * len(shapes), _stream())
                grads.append(gl)
        return (None, None, None, *grads)


class CorrBlock1D:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        corr = _Volume.apply(fmap1, fmap2)  # [B, H, W1, W2]
        b, h, w1, w2 = corr.shape
        self._dims = (b, h, w1)
        levels = [corr]
        for _ in range(self.num_levels):
            levels.append(_Pool.apply(levels[-1]))
        self._levels = levels
        self._sums = _LevelSums(self.num_levels)
        # the reference keeps num_levels+1 entries shaped [B*H*W1, 1, 1, W2 / 2^i]
        self.corr_pyramid = [l.view(b * h * w1, 1, 1, l.shape[-1]) for l in levels]

    def __call__(self, coords):
        return _Lookup.apply(coords, self.radius, self._sums, *self._levels[: self.num_levels])

    @staticmethod
    def corr(fmap1, fmap2):
        c = _Volume.apply(fmap1, fmap2)
        b, h, w1, w2 = c.shape
        return c.view(b, h, w1, 1, w2)


def _not_targeted(name, why):
    class _Stub:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name}: {why}")

    _Stub.__name__ = name
    return _Stub


CorrBlockFast1D = _not_targeted(
    "CorrBlockFast1D", "needs the un-vendored `corr_sampler` CUDA extension (absent from the reference tree)")
AlternateCorrBlock = _not_targeted(
    "AlternateCorrBlock", "raises NotImplementedError in the reference as well (nets/raft/corr.py:166)")
PytorchAlternateCorrBlock1D = _not_targeted(
    "PytorchAlternateCorrBlock1D", "the `alt` variant differs numerically from `reg` (SURVEY.md 2.2) and is "
    "not a parity target; use cfg.MODEL.CORR_IMPLEMENTATION = 'reg'")
