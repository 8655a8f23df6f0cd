"""Autograd wrappers of the 2-D convolution family (include/azhip.h K13): every Conv2d of the feature
extractor (reference nets/psmnet/psmnet_submodule_3.py:13-41, 92-220) and the 2-D convolutions of the
factored cost-volume convolution (costconv.py) on hand-written gfx950 kernels -- forward, input gradient
and weight gradient.  There is no vendor-library path and no fallback: an unsupported geometry raises.

Tensors are NCHW-shaped in torch.channels_last memory, i.e. contiguous [B,H,W,C] rows; weights stay in
PyTorch's [Cout,Cin,KH,KW] layout (reference checkpoints load unchanged), the kernels read a packed bf16
triplet image made per call (memoised between no_grad forwards).

    layer geometry                         forward                     input gradient            weight gradient
    3x3 / 1x1 / 3x5, stride 1, "same"      az_conv2d_fwd               az_conv2d_fwd, flipped    az_conv2d_wgrad
    3x3 d1 with 32/64 -> 32/64 channels    az_conv2d_roll_fwd          the same, flipped         az_conv2d_wgrad
       (firstconv[1..2], layer1, layer2: the batch-walking kernel of az_conv2d_roll.hip; AZ_CONV2D_ROLL=0: off)
    3x3 stride 2, 32/64 channels           3-D gather kernel, mode 1   mode 2 (transposed)       3-D wgrad kernel, stride 2
       (layer2.0.conv1; the image is a depth-1 volume)
    3x3 stride 2 on a 3/6-channel image    az_im2col_s2k3 + 1x1        1x1 flipped + az_col2im   1x1 wgrad on the patches
       (firstconv.0)
    1x1 stride 2 (layer2.0.downsample)     subsample + 1x1             (autograd of the slice)   1x1 wgrad
"""
import os

import torch

from . import _lib, conv3d, overlap, profiler
from .conv3d import _cache_get, _cache_key, _cache_put
from .ops import _call, _chk, _p, _stream

_SAME = {(3, 3, 1), (3, 3, 2), (1, 1, 1), (3, 5, 1)}
_PACK2D_CACHE = {}
PEAK_X6 = (2500.0 / 6.0, "bf16x6: bf16 MFMA peak / 6")


def rows(t):
    """[N,C,H,W] (channels_last memory) -> the same storage as contiguous [N,H,W,C] (a copy otherwise)."""
    return t.permute(0, 2, 3, 1).contiguous()


def image(r):
    """[N,H,W,C] rows -> [N,C,H,W] view in channels_last memory."""
    return r.permute(0, 3, 1, 2)


def _pack(weight, cin, cout, ci_real, co_real, s_out, s_in, kh, kw, flip, cache=False):
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if cache:
        key = (_cache_key(weight), cin, cout, ci_real, co_real, s_out, s_in, kh, kw, bool(flip))
        hit = _cache_get(_PACK2D_CACHE, key)
        if hit is not None:
            return hit[0]
    n = _lib.lib().az_conv2d_packed_floats(cin, cout, kh, kw)
    if n < 0:
        raise RuntimeError(f"conv2d: unsupported channel counts cin={cin} cout={cout}")
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _call("az_conv2d_pack_weights", _p(packed), _p(w), cin, cout, ci_real, co_real, s_out, s_in, kh, kw,
          int(flip), _stream())
    if key is not None:
        _cache_put(_PACK2D_CACHE, key, (packed, weight), 256)
    return packed


PEAK_F16 = conv3d._PEAK_F16


def _amax_of(t, r):
    """device scalar max |t| for the f16x3 kernels: what t's producer attached (BatchNorm apply / backward), else a pass
    over its rows r"""
    am = conv3d._get_amax(t)
    return am if am is not None else conv3d.absmax(r)


def _pack_f16(weight, cin, cout, ci_real, co_real, s_out, s_in, kh, kw, flip, cache=False):
    """(packed f16x3 image, device scalar max |w|) -- az_conv2d_pack_weights_f16"""
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if cache:
        key = (_cache_key(weight), "f16", cin, cout, ci_real, co_real, s_out, s_in, kh, kw, bool(flip))
        hit = _cache_get(_PACK2D_CACHE, key)
        if hit is not None:
            return hit[0]
    if not cache:
        hit = conv3d._planned_pack(weight, w, conv3d.PACK_2D_SAME, cin, cout, ci_real, co_real, s_out, s_in, kh * kw, flip,
                                   lambda packed, w_amax: _call("az_conv2d_pack_weights_f16", _p(packed), _p(w), _p(w_amax), cin,
                                                                cout, ci_real, co_real, s_out, s_in, kh, kw, int(flip), _stream()))
        if hit is not None:
            return hit
    w_amax = _w_amax(weight, w)
    packed = torch.empty(kh * kw * cin * cout, dtype=torch.float32, device=w.device)
    _call("az_conv2d_pack_weights_f16", _p(packed), _p(w), _p(w_amax), cin, cout, ci_real, co_real, s_out, s_in, kh, kw,
          int(flip), _stream())
    if key is not None:
        _cache_put(_PACK2D_CACHE, key, ((packed, w_amax), weight), 256)
    return packed, w_amax


def _run_f16(xr, x_amax, packed, w_amax, cin, cout, kh, kw, dil, scale=None, shift=None, res=None, relu=False,
             tag="conv2d", stats=None):
    """_run / _run_stats on the f16x3 kernels"""
    b, h, w, cx = xr.shape
    out = xr.new_empty(b, h, w, cout)
    with profiler.scope(f"{tag}_{kh}x{kw}d{dil}_{cin}_{cout}", flops=2.0 * kh * kw * cin * cout * b * h * w,
                        peak=PEAK_F16):
        if stats is not None:
            tiles = int(_lib.lib().az_conv2d_stats_tiles(b, h, w, stats.groups))
            part, cnt = xr.new_empty(stats.groups, cout, tiles, 2), xr.new_empty(stats.groups, tiles)
            _call("az_conv2d_fwd_stats_f16", _p(out), _p(part), _p(cnt), _p(xr), _p(packed), _p(x_amax), _p(w_amax),
                  stats.groups, b, h, w, cin, cout, cx, cout, kh, kw, dil, _stream())
            stats.part, stats.cnt, stats.tiles = part, cnt, tiles
        else:
            _call("az_conv2d_fwd_f16", _p(out), _p(xr), _p(packed), _p(x_amax), _p(w_amax), _p(scale), _p(shift), _p(res),
                  int(relu), b, h, w, cin, cout, cx, cout, res.shape[-1] if res is not None else 0, kh, kw, dil, _stream())
    return out


def _up(n, m):
    return (n + m - 1) // m * m


_ROLL2D = os.environ.get("AZ_CONV2D_ROLL", "1") != "0"


def _roll_ok(xr, cin, cout, kh, kw, dil, res=None):
    """the layer is one az_conv2d_roll.hip takes: 3x3, dilation 1, 32 / 64 channels on both sides, dense tensors"""
    return (_ROLL2D and kh == 3 and kw == 3 and dil == 1 and cin in (32, 64) and cout in (32, 64) and xr.shape[-1] == cin
            and (res is None or res.shape[-1] == cout)
            # one statistic group of a tensor goes through a 32-bit buffer offset there (conservatively: the whole batch)
            and xr.shape[0] * xr.shape[1] * xr.shape[2] * 256 < 0xffffff00)


def _pack_roll(weight, cin, cout, s_out, s_in, flip, cache=False):
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if cache:
        key = (_cache_key(weight), "roll", cin, cout, s_out, s_in, bool(flip))
        hit = _cache_get(_PACK2D_CACHE, key)
        if hit is not None:
            return hit[0]
    packed = torch.empty(int(_lib.lib().az_conv2d_roll_packed_floats(cin, cout)), dtype=torch.float32, device=w.device)
    _call("az_conv2d_roll_pack", _p(packed), _p(w), cin, cout, s_out, s_in, int(flip), _stream())
    if key is not None:
        _cache_put(_PACK2D_CACHE, key, (packed, weight), 256)
    return packed


def _w_amax(weight, w):
    """amax array of a weight tensor, once per optimizer step (conv3d._W_AMAX)"""
    wkey = (weight.data_ptr(), weight._version, weight.device.index, weight.numel())
    hit = _cache_get(conv3d._W_AMAX, wkey)
    if hit is not None:
        return hit[0]
    w_amax = conv3d.absmax(w)
    _cache_put(conv3d._W_AMAX, wkey, (w_amax, weight), 512)
    return w_amax


def _pack_roll_f16(weight, cin, cout, s_out, s_in, flip):
    w = _chk(weight.detach().contiguous(), "weight")
    hit = conv3d._planned_pack(weight, w, conv3d.PACK_2D_ROLL, cin, cout, cin, cout, s_out, s_in, 9, flip,
                               lambda packed, w_amax: _call("az_conv2d_roll_pack_f16", _p(packed), _p(w), _p(w_amax), cin, cout,
                                                            s_out, s_in, int(flip), _stream()))
    if hit is not None:
        return hit
    w_amax = _w_amax(weight, w)
    packed = torch.empty(9 * cin * cout, dtype=torch.float32, device=w.device)
    _call("az_conv2d_roll_pack_f16", _p(packed), _p(w), _p(w_amax), cin, cout, s_out, s_in, int(flip), _stream())
    return packed, w_amax


def _run_roll_f16(xr, x_amax, packed, w_amax, cin, cout, res=None, tag="conv2d", stats=None):
    b, h, w, _ = xr.shape
    out = xr.new_empty(b, h, w, cout)
    with profiler.scope(f"{tag}_3x3d1_{cin}_{cout}", flops=2.0 * 9 * cin * cout * b * h * w, peak=PEAK_F16):
        if stats is not None:
            nrows = int(_lib.lib().az_conv2d_roll_stats_rows(stats.groups, b, h, w, cin, cout))
            if nrows <= 0:
                raise RuntimeError(f"az_conv2d_roll_stats_rows: {nrows}")
            part, cnt = xr.new_empty(stats.groups, cout, nrows, 2), xr.new_empty(stats.groups, nrows)
            _call("az_conv2d_roll_fwd_stats_f16", _p(out), _p(part), _p(cnt), _p(xr), _p(packed), _p(x_amax), _p(w_amax),
                  stats.groups, b, h, w, cin, cout, _stream())
            stats.part, stats.cnt, stats.tiles = part, cnt, nrows
        else:
            _call("az_conv2d_roll_fwd_f16", _p(out), _p(xr), _p(packed), _p(x_amax), _p(w_amax), None, None, _p(res), 0,
                  b, h, w, cin, cout, _stream())
    return out


def _run_roll(xr, packed, cin, cout, scale=None, shift=None, res=None, relu=False, tag="conv2d"):
    b, h, w, _ = xr.shape
    out = xr.new_empty(b, h, w, cout)
    with profiler.scope(f"{tag}_3x3d1_{cin}_{cout}", flops=2.0 * 9 * cin * cout * b * h * w, peak=PEAK_X6):
        _call("az_conv2d_roll_fwd", _p(out), _p(xr), _p(packed), _p(scale), _p(shift), _p(res), int(relu), b, h, w,
              cin, cout, _stream())
    return out


def _run_roll_stats(xr, packed, cin, cout, stats):
    b, h, w, _ = xr.shape
    out = xr.new_empty(b, h, w, cout)
    nrows = int(_lib.lib().az_conv2d_roll_stats_rows(stats.groups, b, h, w, cin, cout))
    if nrows <= 0:
        raise RuntimeError(f"az_conv2d_roll_stats_rows: {nrows}")
    part, cnt = xr.new_empty(stats.groups, cout, nrows, 2), xr.new_empty(stats.groups, nrows)
    with profiler.scope(f"conv2d_3x3d1_{cin}_{cout}", flops=2.0 * 9 * cin * cout * b * h * w, peak=PEAK_X6):
        _call("az_conv2d_roll_fwd_stats", _p(out), _p(part), _p(cnt), _p(xr), _p(packed), stats.groups, b, h, w,
              cin, cout, _stream())
    stats.part, stats.cnt, stats.tiles = part, cnt, nrows
    return out


def _run(xr, packed, cin, cout, kh, kw, dil, scale=None, shift=None, res=None, relu=False, tag="conv2d"):
    """xr: [B,H,W,Cx] rows with Cx >= cin; returns [B,H,W,cout] rows."""
    b, h, w, cx = xr.shape
    out = xr.new_empty(b, h, w, cout)
    with profiler.scope(f"{tag}_{kh}x{kw}d{dil}_{cin}_{cout}", flops=2.0 * kh * kw * cin * cout * b * h * w,
                        peak=PEAK_X6):
        _call("az_conv2d_fwd", _p(out), _p(xr), _p(packed), _p(scale), _p(shift), _p(res), int(relu), b, h, w,
              cin, cout, cx, cout, res.shape[-1] if res is not None else 0, kh, kw, dil, _stream())
    return out


def _run_stats(xr, packed, cin, cout, kh, kw, dil, stats):
    """_run without epilogue operands; also fills `stats` (bn2d.Partials) with the output's BatchNorm partials."""
    b, h, w, cx = xr.shape
    out = xr.new_empty(b, h, w, cout)
    tiles = int(_lib.lib().az_conv2d_stats_tiles(b, h, w, stats.groups))
    part, cnt = xr.new_empty(stats.groups, cout, tiles, 2), xr.new_empty(stats.groups, tiles)
    with profiler.scope(f"conv2d_{kh}x{kw}d{dil}_{cin}_{cout}", flops=2.0 * kh * kw * cin * cout * b * h * w,
                        peak=PEAK_X6):
        _call("az_conv2d_fwd_stats", _p(out), _p(part), _p(cnt), _p(xr), _p(packed), stats.groups, b, h, w, cin, cout,
              cx, cout, kh, kw, dil, _stream())
    stats.part, stats.cnt, stats.tiles = part, cnt, tiles
    return out


def _wgrad(gr, xr, cm, cn, cm_real, cn_real, kh, kw, dil, tag="conv2d", sink=None, amax=None, late_ok=True):
    """gr: [B,H,W,>=cm] grad rows, xr: [B,H,W,>=cn] input rows -> [cm_real, cn_real, kh, kw].
    sink: overlap.Sink or None -- the kernels run on its side stream (the result is valid after its join).
    amax: None (bf16x6) or (amax array of gr or None, of xr or None): the f16x3 kernels; a missing one is taken here.
    late_ok = False: the caller reads the result on the side stream right away: no deferred epilogue (overlap.Sink)."""
    b, h, w, _ = xr.shape
    gw = xr.new_empty(cm_real, cn_real, kh, kw)
    ws_bytes = _lib.lib().az_conv2d_wgrad_workspace(cm, cn, kh, kw)
    if ws_bytes < 0:
        raise RuntimeError(f"conv2d wgrad: unsupported channel counts {cm} x {cn}")
    keep = [t for t in (amax or ()) if t is not None]
    with overlap.scope(sink, gr, xr, gw, *keep):  # (gw too: the engine may drop it before the join, overlap.py)
        # on the side stream the epilogue is deferred: zeroed workspace from the pass's arena, unpack at the join (overlap.Sink)
        late = late_ok and sink is not None and sink.live and overlap.DEFER_UNPACK
        ws = sink.take_workspace(ws_bytes // 4) if late else xr.new_empty(ws_bytes // 4)
        out = None if late else _p(gw)
        if amax is None:
            with profiler.scope(f"{tag}_wgrad_{kh}x{kw}d{dil}_{cm}_{cn}", flops=2.0 * kh * kw * cm * cn * b * h * w,
                                peak=PEAK_X6):
                _call("az_conv2d_wgrad", out, _p(ws), ws_bytes, _p(gr), _p(xr), b, h, w, cm, cn, cm_real, cn_real,
                      gr.shape[-1], xr.shape[-1], kh, kw, dil, _stream())
        else:
            am_g = amax[0] if amax[0] is not None else conv3d.absmax(gr)
            am_x = amax[1] if amax[1] is not None else conv3d.absmax(xr)
            with profiler.scope(f"{tag}_wgrad_{kh}x{kw}d{dil}_{cm}_{cn}", flops=2.0 * kh * kw * cm * cn * b * h * w,
                                peak=PEAK_F16):
                _call("az_conv2d_wgrad_f16", out, _p(ws), ws_bytes, _p(gr), _p(xr), _p(am_g), _p(am_x), b, h, w, cm, cn,
                      cm_real, cn_real, gr.shape[-1], xr.shape[-1], kh, kw, dil, _stream())
            if sink is not None and sink.live:
                sink.keep.extend((am_g, am_x))
        if late:
            sink.defer_unpack(gw, ws, cm, cn, cm_real, cn_real, kh * kw)
    return gw


_OVERLAP_2D = os.environ.get("AZ_2D_WGRAD_OVERLAP", "1") != "0"  # (scheduling experiment, tools/sched_ab.sh)


def _leaf_sink(sink, weight):
    if not _OVERLAP_2D:
        return None
    """A sink lets a weight gradient be handed to autograd before its kernel has run; that is only sound when
    the next reader is the pass's own gate node (overlap.py), i.e. when `weight` is one of the sink's gated
    parameters -- not for derived tensors such as the merged kernels of the cost-volume convolution."""
    return sink if (sink is not None and sink.owns(weight)) else None


def _w(m, arith):
    """the weight tensor a layer uses in this pass: the sink's gated alias, or the parameter itself"""
    return arith.sink.weight(m.weight) if arith.sink is not None else m.weight


class _ConvSame(torch.autograd.Function):
    """conv2d(x, w, stride 1, "same" padding, dilation) for the geometries of _SAME.

    with_skip: also return x itself as a second output.  A residual block hands that alias to its shortcut
    (psmnet_submodule_3.py:69-78: out = conv2(conv1(x)) + x), so both gradients of x arrive HERE and the
    shortcut's is added in the input-gradient convolution's epilogue instead of by a separate elementwise
    kernel of the autograd engine (one read + one write of the activation per block saved)."""

    @staticmethod
    def forward(ctx, x, weight, dil, with_skip, sink=None, stats=None, f16=False):
        ctx.sink = _leaf_sink(sink, weight)
        ctx.f16 = f16
        cout, cin, kh, kw = weight.shape
        xr = _chk(rows(x), "x")
        with torch.cuda.device(x.device):
            want_stats = stats is not None and (kh, kw) in ((3, 3), (1, 1)) and xr.shape[0] % stats.groups == 0
            if f16 and _roll_ok(xr, cin, cout, kh, kw, dil):
                pk, w_amax = _pack_roll_f16(weight, cin, cout, cin * 9, 9, False)
                y = _run_roll_f16(xr, _amax_of(x, xr), pk, w_amax, cin, cout, stats=stats if want_stats else None)
            elif f16:
                pk, w_amax = _pack_f16(weight, cin, cout, cin, cout, cin * kh * kw, kh * kw, kh, kw, False)
                y = _run_f16(xr, _amax_of(x, xr), pk, w_amax, cin, cout, kh, kw, dil, stats=stats if want_stats else None)
            elif _roll_ok(xr, cin, cout, kh, kw, dil):
                pk = _pack_roll(weight, cin, cout, cin * 9, 9, False)
                y = _run_roll_stats(xr, pk, cin, cout, stats) if want_stats else _run_roll(xr, pk, cin, cout)
            else:
                pk = _pack(weight, cin, cout, cin, cout, cin * kh * kw, kh * kw, kh, kw, False)
                if want_stats:
                    y = _run_stats(xr, pk, cin, cout, kh, kw, dil, stats)  # + the BatchNorm partials of y (bn2d.Partials)
                else:
                    y = _run(xr, pk, cin, cout, kh, kw, dil)
        ctx.save_for_backward(xr, weight)
        ctx.dil = dil
        ctx.x_amax = (conv3d._get_amax(x) if conv3d._get_amax(x) is not None else conv3d._get_amax(xr)) if f16 else None
        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None, not as a zero tensor
        if with_skip:
            return image(y), x.view_as(x)
        return image(y)

    @staticmethod
    def backward(ctx, gy, gskip=None):
        xr, weight = ctx.saved_tensors
        cout, cin, kh, kw = weight.shape
        dil = ctx.dil
        if gy is None:  # only the shortcut was used downstream
            return gskip, None, None, None, None, None, None
        gr = _chk(rows(gy), "grad_y")
        gx = gw = None
        with torch.cuda.device(gy.device):
            if ctx.needs_input_grad[0]:  # the same convolution, taps flipped, channel roles swapped
                sk = _chk(rows(gskip), "grad_skip") if gskip is not None else None
                if ctx.f16 and _roll_ok(gr, cout, cin, kh, kw, dil, sk):
                    pk, w_amax = _pack_roll_f16(weight, cout, cin, 9, cin * 9, True)
                    gx = image(_run_roll_f16(gr, _amax_of(gy, gr), pk, w_amax, cout, cin, res=sk, tag="dgrad2d"))
                elif ctx.f16:
                    pk, w_amax = _pack_f16(weight, cout, cin, cout, cin, kh * kw, cin * kh * kw, kh, kw, True)
                    gx = image(_run_f16(gr, _amax_of(gy, gr), pk, w_amax, cout, cin, kh, kw, dil, res=sk, tag="dgrad2d"))
                elif _roll_ok(gr, cout, cin, kh, kw, dil, sk):
                    pk = _pack_roll(weight, cout, cin, 9, cin * 9, True)
                    gx = image(_run_roll(gr, pk, cout, cin, res=sk, tag="dgrad2d"))
                else:
                    pk = _pack(weight, cout, cin, cout, cin, kh * kw, cin * kh * kw, kh, kw, True)
                    gx = image(_run(gr, pk, cout, cin, kh, kw, dil, res=sk, tag="dgrad2d"))
            if ctx.needs_input_grad[1]:
                am = None
                if ctx.f16:
                    g_am = conv3d._get_amax(gy)
                    am = (g_am if g_am is not None else conv3d._get_amax(gr), ctx.x_amax)
                gw = _wgrad(gr, xr, cout, cin, cout, cin, kh, kw, dil, sink=ctx.sink, amax=am)
        return gx, gw, None, None, None, None, None


def _check_same(weight, dilation):
    cout, cin, kh, kw = weight.shape
    if (kh, kw, dilation if kh > 1 else 1) not in _SAME or cin % 32 or cout % 32:
        raise RuntimeError(f"conv_same: unsupported geometry {tuple(weight.shape)}, dilation {dilation}")
    return 1 if kh == 1 else dilation


def conv_same(x, weight, dilation=1, sink=None, stats=None, f16=False):
    """F.conv2d(x, weight, padding="same", dilation=dilation) for [B,C,H,W] x (channels_last preferred).
    stats: a bn2d.Partials to fill with the BatchNorm partials of the output (3x3 and 1x1 layers).
    f16: forward and input gradient on the f16x3 arithmetic (conv3d.F16X3) where a kernel for the shape exists."""
    return _ConvSame.apply(x, weight, _check_same(weight, dilation), False, sink, stats, f16)


def conv_same_skip(x, weight, dilation=1, sink=None, stats=None, f16=False):
    """(conv_same(x, weight, dilation), x): the second output is x for the block's shortcut (see _ConvSame)."""
    return _ConvSame.apply(x, weight, _check_same(weight, dilation), True, sink, stats, f16)


class _ConvS2Vol(torch.autograd.Function):
    """3x3, stride 2, pad 1 with 32/64 channels (layer2.0.conv1): the image as a depth-1 volume on the 3-D
    gather kernels -- forward = stride-2 mode, input gradient = transposed mode (its second output
    plane, the kd = 2 taps, is discarded), weight gradient = the stride-2 3-D wgrad kernel's centre slice."""

    @staticmethod
    def forward(ctx, x, weight, arith):
        cout, cin = weight.shape[:2]
        xv = _chk(rows(x).unsqueeze(1), "x")
        w3 = weight.detach().new_zeros(cout, cin, 3, 3, 3)
        w3[:, :, 1] = weight.detach()
        with torch.cuda.device(x.device):
            y = conv3d._conv(xv, w3, conv3d.CONV_S2, arith.conv, tag="fe2d_s2")
        ctx.save_for_backward(xv, w3)
        ctx.arith = arith
        ctx.sink = _leaf_sink(arith.sink, weight)
        return image(y.squeeze(1))

    @staticmethod
    def backward(ctx, gy):
        xv, w3 = ctx.saved_tensors
        cout, cin = w3.shape[:2]
        arith = ctx.arith
        gv = _chk(rows(gy).unsqueeze(1), "grad_y")
        gx = gw = None
        with torch.cuda.device(gy.device):
            if ctx.needs_input_grad[0]:
                g2 = conv3d._input_grad(gv, w3, conv3d.CONV_S2, cin, cout, arith.conv)  # [B,2,H,W,cin]
                gx = image(g2[:, 0, :xv.shape[2], :xv.shape[3]])
            if ctx.needs_input_grad[1]:
                g3 = conv3d._weight_grad(xv, gv, conv3d.CONV_S2, cin, cout, arith.wgrad, ctx.sink, late_ok=False)  # (sliced right below)
                gw = g3.new_empty(cout, cin, 3, 3)
                with overlap.scope(ctx.sink, g3, gw):  # the slice reads g3 on the stream that wrote it
                    gw.copy_(g3[:, :, 1])
        return gx, gw, None


class _ConvS2Patches(torch.autograd.Function):
    """3x3, stride 2, pad 1 on a thin image (C = 3 or 6 -> 32): patch extraction + 1x1 convolution."""

    @staticmethod
    def forward(ctx, x, weight, sink=None, token=None, stats=None):
        # `token`: the sink's 1-element tensor (overlap._Tail's output).  This layer is the first convolution of
        # the extractor, hence the last convolution node of the backward pass: its gradient for the token is
        # what makes _Tail.backward -- the join of the side stream -- run after every weight-gradient launch.
        ctx.sink = _leaf_sink(sink, weight)
        if ctx.sink is not None and token is not None and token.requires_grad:
            ctx.sink.armed = True
        cout, cin = weight.shape[:2]
        xr = _chk(rows(x), "x")
        b, h, w, _ = xr.shape
        kp = _up(9 * cin, 32)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        patches = xr.new_empty(b, ho, wo, kp)
        # [cout, cin, 3, 3] -> [cout, (tap, cin)] as the patches are laid out
        w2 = weight.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
        with torch.cuda.device(x.device):
            _call("az_im2col_s2k3", _p(patches), _p(xr), b, cin, h, w, kp, _stream())
            pk = _pack(w2, kp, cout, 9 * cin, cout, 9 * cin, 1, 1, 1, False)
            if stats is not None and b % stats.groups == 0:
                y = _run_stats(patches, pk, kp, cout, 1, 1, 1, stats)
            else:
                y = _run(patches, pk, kp, cout, 1, 1, 1, tag="fe2d_first")
        ctx.save_for_backward(patches, w2)
        ctx.dims = (b, cin, h, w, kp)
        return image(y)

    @staticmethod
    def backward(ctx, gy):
        patches, w2 = ctx.saved_tensors
        b, cin, h, w, kp = ctx.dims
        cout = w2.shape[0]
        gr = _chk(rows(gy), "grad_y")
        gx = gw = None
        with torch.cuda.device(gy.device):
            if ctx.needs_input_grad[0]:
                pk = _pack(w2, cout, kp, cout, 9 * cin, 1, 9 * cin, 1, 1, True)
                gp = _run(gr, pk, cout, kp, 1, 1, 1, tag="dgrad2d_first")
                gxr = gr.new_empty(b, h, w, cin)
                _call("az_col2im_s2k3", _p(gxr), _p(gp), b, cin, h, w, kp, _stream())
                gx = image(gxr)
            if ctx.needs_input_grad[1]:
                g2 = _wgrad(gr, patches, cout, kp, cout, 9 * cin, 1, 1, 1, tag="fe2d_first", sink=ctx.sink, late_ok=False)  # [cout, 9cin,1,1]; permuted right below
                gw = g2.new_empty(cout, cin, 3, 3)
                with overlap.scope(ctx.sink, g2, gw):
                    gw.copy_(g2.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2))
        gtok = gr.new_zeros(1) if ctx.needs_input_grad[3] else None
        return gx, gw, None, gtok, None


def is_same(m):
    """True for the stride-1 "same"-padded layers (the route of conv_same / conv_same_skip)."""
    k, s, d, p = m.kernel_size, m.stride, m.dilation, m.padding
    return s == (1, 1) and k[0] == k[1] and p[0] == p[1] == d[0] * (k[0] - 1) // 2 and d[0] == d[1]


def conv(x, m, arith=None, skip=False, stats=None):
    """m(x) for an nn.Conv2d of the extractor (bias-free, groups 1), differentiable, on the HIP kernels.
    skip=True (stride-1 layers only): returns (m(x), x), see conv_same_skip.
    stats: a bn2d.Partials; the stride-1 3x3 / 1x1 routes fill it with the output's BatchNorm partials (the other
    routes leave it empty and the BatchNorm runs its own statistics pass)."""
    arith = conv3d._arith(arith)
    if not isinstance(m, torch.nn.Conv2d) or m.bias is not None or m.groups != 1:
        raise RuntimeError("conv2d.conv: expects a bias-free nn.Conv2d")
    k, s, d, p = m.kernel_size, m.stride, m.dilation, m.padding
    cin, cout = m.in_channels, m.out_channels
    if is_same(m):
        return (conv_same_skip if skip else conv_same)(x, _w(m, arith), d[0], arith.sink, stats,
                                                       arith.conv == conv3d.F16X3)
    if skip:
        raise RuntimeError(f"conv2d.conv: skip output needs a stride-1 layer, got {m}")
    if s == (2, 2) and k == (3, 3) and p == (1, 1) and d == (1, 1):
        if cin in (32, 64) and cout in (32, 64) and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0:
            return _ConvS2Vol.apply(x, _w(m, arith), arith)
        if 9 * cin <= 64 and cout % 32 == 0:
            sink = arith.sink
            return _ConvS2Patches.apply(x, _w(m, arith), sink, sink.token if sink is not None else None, stats)
    if s == (2, 2) and k == (1, 1) and p == (0, 0):
        return conv_same(x[:, :, ::2, ::2].contiguous(memory_format=torch.channels_last), _w(m, arith), 1, arith.sink, stats,
                         arith.conv == conv3d.F16X3)
    raise RuntimeError(f"conv2d.conv: unsupported layer {m}")


def conv_bn_eval(x, m, bn, relu=False, residual=None):
    """Inference: relu?(BatchNorm2d(conv(x)) + residual) with the running-statistics affine map, the
    residual sum and the ReLU folded into the convolution's epilogue (one pass, no normalisation
    kernel).  Only for the stride-1 geometries; returns None when the layer needs the generic route."""
    k, s, d, p = m.kernel_size, m.stride, m.dilation, m.padding
    cin, cout = m.in_channels, m.out_channels
    if not (s == (1, 1) and k[0] == k[1] and p[0] == p[1] == d[0] * (k[0] - 1) // 2 and cin % 32 == 0
            and cout % 32 == 0 and (k[0], k[1], d[0] if k[0] > 1 else 1) in _SAME):
        return None
    kh, kw = k
    xr = _chk(rows(x), "x")
    rr = _chk(rows(residual), "residual") if residual is not None else None
    with torch.cuda.device(x.device):
        scale, shift = conv3d.eval_affine(bn, xr, cache=True)
        # (NOT on the batch-walking kernel: v_mfma_f32_16x16x32_bf16 sums 32 products per instruction and its signed mean
        #  error is -9e-9 of mean|y| against -6e-9 for the 32x32x16 form (tools/conv2d_bias_probe.py; DESIGN.md section 3
        #  on where that floor comes from).  Train-mode BatchNorm subtracts it with the batch mean; with the
        #  running-statistics map of inference it survives the 64x64 SPP averages, and the full-size eval output moved
        #  from 7.0e-4 to 1.05e-3 px max error against the exact result -- over the 1e-3 bar -- for 16 % less latency.)
        pk = _pack(m.weight, cin, cout, cin, cout, cin * kh * kw, kh * kw, kh, kw, False, cache=True)
        return image(_run(xr, pk, cin, cout, kh, kw, d[0] if kh > 1 else 1, scale, shift, rr, relu, tag="conv2d_eval"))
