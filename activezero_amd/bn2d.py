"""BatchNorm2d (+ residual) (+ ReLU) of the 2-D feature extractor on the HIP BatchNorm kernels
(csrc/az_bn3d.hip), for the channels-last outputs of the 2-D convolution kernels (conv2d.py).

`groups` splits the batch into equal consecutive parts that get their OWN batch statistics and
update the running statistics one after the other.  With groups=2 one pass over the stacked
(left, right) batch is exactly the reference's two sequential calls of `feature_extraction`
(nets/psmnet/psmnet_3.py:145-146) -- same per-call statistics, same running-stat updates -- while
every convolution, residual sum and weight gradient runs once on 2B images.
"""
import torch

from . import _lib
from .ops import _call, _chk, _p, _stream

CHANNELS = (32, 64, 128)


def supported(bn, x):
    return (isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and bn.affine and bn.num_features in CHANNELS and x.is_cuda
            and x.dtype == torch.float32 and (bn.momentum is not None or not bn.track_running_stats))


def _rows(t):
    """[N,C,H,W] in channels_last memory -> the same storage as contiguous [N,H,W,C] (no copy when the
    tensor already is channels_last)."""
    return t.permute(0, 2, 3, 1).contiguous()


class Partials:
    """BatchNorm partials of a tensor made by its producing convolution (conv2d.conv(..., stats=...)): filled by
    the convolution's forward, consumed by bn_act in place of a statistics pass over the tensor."""

    def __init__(self, groups):
        self.groups, self.part, self.cnt, self.tiles = groups, None, None, 0

    @property
    def ready(self):
        return self.part is not None


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, groups, cache, partials=None):
        n, c, h, w = x.shape
        if n % groups:
            raise RuntimeError("batch must divide into the statistic groups")
        xr = _chk(_rows(x), "x")
        rr = _chk(_rows(residual), "residual") if residual is not None else None
        yr = torch.empty_like(xr)
        nvox = (n // groups) * h * w
        training = bn.training or not bn.track_running_stats
        eps = float(bn.eps)
        g_, b_ = gamma.detach(), beta.detach()
        lib = _lib.lib()
        with torch.cuda.device(x.device):
            if not training:  # running statistics: one affine map for the whole batch
                from .conv3d import eval_affine
                scale, shift = eval_affine(bn, xr, cache=cache)
                _call("az_bn3d_apply", _p(yr), _p(xr), _p(scale), _p(shift), _p(rr), int(relu), nvox * groups, c,
                      None, _stream())
                ctx.save_for_backward(xr, yr if relu else None, gamma, scale, bn.running_mean.clone())
                ctx.cfg = (False, relu, residual is not None, groups, (n, c, h, w))
                return yr.permute(0, 3, 1, 2)
            ws_bytes = lib.az_bn2d_workspace(groups, nvox, c)
            if ws_bytes < 0:
                _lib.check(int(ws_bytes), "az_bn2d_workspace")
            ws = xr.new_empty(ws_bytes // 4)
            stats = xr.new_empty(4, groups, c)  # mean, invstd, scale, shift per group
            track = bn.track_running_stats and bn.running_mean is not None
            nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
            pre = partials is not None and partials.ready and partials.groups == groups
            from . import conv3d as _c3
            y_amax = _c3._ZEROS.take(xr)  # max |y|, taken by the apply kernel: the f16x3 scale of the layers that read y
            _call("az_bn2d_fwd", _p(yr), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
                  _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None, _p(xr), _p(rr),
                  _p(g_), _p(b_), _p(ws), ws_bytes, int(relu), groups, nvox, c, eps,
                  float(bn.momentum) if bn.momentum is not None else 0.1, _p(nbt),
                  _p(partials.part) if pre else None, _p(partials.cnt) if pre else None, partials.tiles if pre else 0,
                  _p(y_amax), _stream())
            if nbt is not None:
                from .conv3d import _touched
                _touched(nbt, bn.running_mean, bn.running_var)
        # a ReLU layer without residual recomputes its mask from x in backward: y need not be kept
        ctx.save_for_backward(xr, yr if (relu and residual is not None) else None, gamma, stats)
        ctx.cfg = (True, relu, residual is not None, groups, (n, c, h, w))
        y = yr.permute(0, 3, 1, 2)  # [N,C,H,W] view in channels_last memory
        _c3._set_amax(y, y_amax)
        return y

    @staticmethod
    def backward(ctx, gy):
        training, relu, has_res, groups, (n, c, h, w) = ctx.cfg
        gr = _chk(_rows(gy), "grad_y")
        if not training:
            # frozen statistics (fine-tuning in eval mode): y = relu?(x*s + t + res), s = gamma*rinv,
            # t = beta - running_mean*s.  A rarely used path: plain tensor ops.
            xr, yr, gamma, scale, rm = ctx.saved_tensors
            dz = gr * (yr > 0).to(gr.dtype) if relu else gr
            g_ = gamma.detach()
            rinv = torch.where(g_ != 0, scale / g_, torch.zeros_like(scale))
            dgamma = (dz * (xr - rm)).sum(dim=(0, 1, 2)) * rinv
            dbeta = dz.sum(dim=(0, 1, 2))
            g_res = dz.permute(0, 3, 1, 2) if has_res else None
            return (dz * scale).permute(0, 3, 1, 2), dgamma, dbeta, g_res, None, None, None, None, None
        xr, yr, gamma, stats = ctx.saved_tensors
        nvox = (n // groups) * h * w
        lib = _lib.lib()
        dxr = torch.empty_like(xr)
        dzr = torch.empty_like(xr) if (has_res and relu) else None
        dgb = gr.new_empty(2, c)
        with torch.cuda.device(gr.device):
            ws_bytes = lib.az_bn2d_workspace(groups, nvox, c)
            ws = gr.new_empty(ws_bytes // 4)
            remask = relu and not has_res
            from . import conv3d as _c3
            dx_amax = _c3._ZEROS.take(gr)
            _call("az_bn2d_bwd", _p(dxr), _p(dzr), _p(dgb[0]), _p(dgb[1]), _p(ws), ws_bytes, _p(gr),
                  _p(yr) if (relu and has_res) else None, _p(xr), _p(stats[0]), _p(stats[1]), _p(gamma.detach()),
                  _p(stats[2]) if remask else None, _p(stats[3]) if remask else None, int(relu), groups, nvox, c,
                  _p(dx_amax), _stream())
        g_res = None
        if has_res:
            g_res = (dzr if relu else gr).permute(0, 3, 1, 2)
        dx = dxr.permute(0, 3, 1, 2)
        _c3._set_amax(dx, dx_amax)
        return dx, dgb[0], dgb[1], g_res, None, None, None, None, None


def bn_act(x, bn, relu=False, residual=None, groups=1, partials=None):
    """relu?(BatchNorm2d(x) + residual), x [N,C,H,W] (channels_last preferred), statistics per batch group.
    partials: a filled `Partials` of x from its producing convolution (train mode: no statistics pass here)."""
    return _BNAct.apply(x, bn.weight, bn.bias, residual, bn, relu, groups, not torch.is_grad_enabled(), partials)
