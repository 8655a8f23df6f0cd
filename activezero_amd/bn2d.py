"""BatchNorm2d (+ residual) (+ ReLU) of the 2-D feature extractor on the HIP BatchNorm kernels
(csrc/az_bn3d.hip), for channels-last activations whose convolution ran elsewhere (MIOpen).

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
    return (isinstance(bn, torch.nn.BatchNorm2d) and bn.affine and bn.num_features in CHANNELS and x.is_cuda
            and x.dtype == torch.float32 and (bn.momentum is not None or not bn.track_running_stats))


def _rows(t):
    """[N,C,H,W] in channels_last memory -> the same storage as contiguous [N,H,W,C] (no copy when the
    tensor already is channels_last)."""
    return t.permute(0, 2, 3, 1).contiguous()


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, groups):
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
        means, invstds = [], []
        with torch.cuda.device(x.device):
            scale, shift = xr.new_empty(c), xr.new_empty(c)
            if not training:
                _call("az_bn3d_eval_affine", _p(scale), _p(shift), _p(g_), _p(b_), _p(bn.running_mean),
                      _p(bn.running_var), eps, c, _stream())
            else:
                tiles = lib.az_bn3d_stats_tiles(nvox, c)
                if tiles < 0:
                    _lib.check(int(tiles), "az_bn3d_stats_tiles")
                part, cnt = xr.new_empty(c, tiles, 2), xr.new_empty(tiles)
                track = bn.track_running_stats and bn.running_mean is not None
            for g in range(groups):
                xs, ys = xr[g * (n // groups):(g + 1) * (n // groups)], yr[g * (n // groups):(g + 1) * (n // groups)]
                rs = rr[g * (n // groups):(g + 1) * (n // groups)] if rr is not None else None
                if training:
                    mean, invstd = xr.new_empty(c), xr.new_empty(c)
                    _call("az_bn3d_stats", _p(part), _p(cnt), _p(xs), nvox, c, _stream())
                    _call("az_bn3d_finalize", _p(mean), _p(invstd), _p(scale), _p(shift),
                          _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                          _p(part), _p(cnt), _p(g_), _p(b_), tiles, c, eps,
                          float(bn.momentum) if bn.momentum is not None else 0.1, _stream())
                    means.append(mean)
                    invstds.append(invstd)
                _call("az_bn3d_apply", _p(ys), _p(xs), _p(scale), _p(shift), _p(rs), int(relu), nvox, c, _stream())
            if training and track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(groups)
        if training:
            ctx.save_for_backward(xr, yr if relu else None, gamma, *means, *invstds)
        ctx.cfg = (training, relu, residual is not None, groups, (n, c, h, w))
        return yr.permute(0, 3, 1, 2)  # [N,C,H,W] view in channels_last memory

    @staticmethod
    def backward(ctx, gy):
        training, relu, has_res, groups, (n, c, h, w) = ctx.cfg
        if not training:
            raise NotImplementedError("eval-mode BatchNorm backward is not implemented on the HIP path; "
                                      "run validation under torch.no_grad() as the reference does")
        xr, yr, gamma = ctx.saved_tensors[:3]
        means, invstds = ctx.saved_tensors[3:3 + groups], ctx.saved_tensors[3 + groups:]
        gr = _chk(_rows(gy), "grad_y")
        nb = n // groups
        nvox = nb * h * w
        lib = _lib.lib()
        dxr = torch.empty_like(xr)
        dzr = torch.empty_like(xr) if (has_res and relu) else None
        dgb = gr.new_empty(groups, 2, c)  # per-group (dgamma, dbeta), summed once below
        with torch.cuda.device(gr.device):
            ws_bytes = lib.az_bn3d_bwd_workspace(nvox, c)
            ws, coef = gr.new_empty(ws_bytes // 4), gr.new_empty(c, 3)
            for g in range(groups):
                sl = slice(g * nb, (g + 1) * nb)
                _call("az_bn3d_bwd", _p(dxr[sl]), _p(dzr[sl]) if dzr is not None else None, _p(dgb[g, 0]),
                      _p(dgb[g, 1]), _p(coef), _p(ws), ws_bytes, _p(gr[sl]), _p(yr[sl]) if relu else None,
                      _p(xr[sl]), _p(means[g]), _p(invstds[g]), _p(gamma.detach()), int(relu), nvox, c, _stream())
        dgb = dgb.sum(0) if groups > 1 else dgb[0]
        dgamma, dbeta = dgb[0], dgb[1]
        g_res = None
        if has_res:
            g_res = (dzr if relu else gr).permute(0, 3, 1, 2)
        return dxr.permute(0, 3, 1, 2), dgamma, dbeta, g_res, None, None, None


def bn_act(x, bn, relu=False, residual=None, groups=1):
    """relu?(BatchNorm2d(x) + residual), x [N,C,H,W] (channels_last preferred), statistics per batch group."""
    return _BNAct.apply(x, bn.weight, bn.bias, residual, bn, relu, groups)
