"""Autograd wrappers of the MFMA 3-D convolution family (include/azhip.h K4/K5).

Volumes are channels-last tensors [B, D, H, W, C] (C in {32, 64}); weights stay in
PyTorch's native layouts ([Cout,Cin,3,3,3] for Conv3d, [Cin,Cout,3,3,3] for
ConvTranspose3d) so the reference state-dict loads unchanged -- the kernels read a
packed copy made per call by az_conv3d_pack_weights (110-442 KB, L2 resident).

Every layer's three passes map onto the same gather kernel (mode 0/1/2 = stride-1
conv, stride-2 conv, stride-2 transposed conv) with differently packed weights:

    layer                 forward      input gradient            weight gradient
    Conv3d stride 1       mode 0       mode 0, flipped+swapped   wgrad(dy, x, 1)
    Conv3d stride 2       mode 1       mode 2, swapped           wgrad(dy, x, 2)
    ConvTranspose3d s2    mode 2       mode 1, as stored         wgrad(x, dy, 2)
"""
import collections
import os
import threading

import torch

from . import _lib, overlap, profiler
from .ops import _call, _chk, _p, _stream

CONV_S1, CONV_S2, DECONV_S2 = 0, 1, 2
FP32, BF16X6, F16X3 = 0, 1, 3
_PREC = {"fp32": FP32, "bf16x6": BF16X6, "f16x3": F16X3}
# "f16x3" (precision 3): operands scaled by a power of two from the tensor's largest magnitude and split into two fp16
# parts, three MFMAs per product, fp32 accumulation (include/azhip.h, az_roll_common.h; per-layer error against fp64 at
# bf16x6's level and below torch's fp32 convolution, tools/f16x3_probe.py).  It is used wherever an f16x3 kernel exists
# for the layer's shape (the _f16_*_ok predicates below); other shapes run on bf16x6.  As the arithmetic of the BACKWARD
# matrix kernels it is on by default with a bf16x6 forward; AZ_BWD_F16=0 (read once) keeps every gradient on the
# forward's arithmetic, for A/B runs.
_BWD_F16 = os.environ.get("AZ_BWD_F16", "1") != "0"
# Pre-split gradients (round 5; include/azhip.h "S2 format"): the BatchNorm-backward apply pass writes d(raw) as the two
# scaled fp16 parts its two readers -- the layer's f16x3 input- and weight-gradient kernels -- would otherwise compute
# while staging, wherever BOTH of them take such an operand (az_conv3d_*_split_ok).  AZ_PRESPLIT=0 (read once): fp32
# gradients everywhere, split by the consumers (round 4), for A/B runs.
PRESPLIT = os.environ.get("AZ_PRESPLIT", "1") != "0"
# include/azhip.h precision 2: the bf16x6 arithmetic on the depth-rolling 16x16x32 kernel (az_conv3d_roll.hip),
# which reads its own packed-weight layout.  Not an arithmetic a caller chooses: _layout() routes the stride-1
# layers with 32 output channels (the V0 layers and their input gradients) there.  AZ_CONV_ROLL=0 (read once)
# keeps them on az_conv3d_m128.hip for A/B runs.
BF16X6_R16 = 2
_ROLL = os.environ.get("AZ_CONV_ROLL", "1") != "0"


_OFF32 = 0xffffff00  # the depth-rolling kernels address ONE batch element through a 32-bit buffer offset


def _fits32(x, cin, cout):
    """the volume x ([B,D,H,W,C] tensor or LazyCostVolume) fits the 32-bit offsets of the depth-rolling kernels"""
    if isinstance(x, LazyCostVolume):
        return True
    _, d, h, w, _ = x.shape
    return d * h * w * max(cin, cout) * 4 < _OFF32


def _fits32_transposed(x, op_cin, op_cout):
    """transposed 64 -> 32 launches (the depth-rolling kernel of az_conv3d_t2roll.hip, 32-bit offsets): both the coarse
    input and the eight-times-larger fine output of a batch element must stay below 4 GiB"""
    if not (op_cin == 64 and op_cout == 32):
        return True
    _, d, h, w, _ = x.shape
    return d * h * w * 64 * 4 < _OFF32 and 8 * d * h * w * 32 * 4 < _OFF32


def _on_s2roll(op_mode, op_cin, op_cout):
    """the stride-2 32 -> 64 operation (hourglass conv1 forward, conv6 input gradient) runs on the depth-rolling kernel of
    az_conv3d_s2roll.hip, which addresses a batch element through 32-bit offsets (AZ_CONV_S2ROLL=0: the gather kernel)"""
    return op_mode == CONV_S2 and op_cin == 32 and op_cout == 64 and _lib.lib().az_option(b"AZ_CONV_S2ROLL") != 0


def _layout(precision, mode, op_cout, fits32=True):
    """precision code the C ABI is called with (packing and launch must agree)"""
    if precision == BF16X6 and mode == CONV_S1 and op_cout == 32 and _ROLL and fits32:
        return BF16X6_R16
    return precision


class Arith(collections.namedtuple("Arith", "conv wgrad sink bwd16", defaults=(None, False))):
    """Arithmetic of the MFMA kernels, passed with every call (no process-wide switch):
      fp32   -- v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chain (157 TFLOP/s peak)
      bf16x6 -- exact 3-way bf16 split of both operands, six bf16 MFMAs per product, fp32
                accumulate: fp32-class accuracy (measured ~1e-7 relative) at 2.7x the rate
    `conv` covers forward / input-gradient / transposed kernels, `wgrad` the weight gradients.
    `sink` (overlap.Sink or None) is not arithmetic but travels the same way, with every call: when set, the
    weight-gradient kernels of this forward/backward pass run on its side stream (overlap.py).
    `bwd16`: input and weight gradients on the f16x3 kernels where one exists for the shape (default with bf16x6;
    never with fp32, the exact mode)."""
    __slots__ = ()

    @classmethod
    def of(cls, conv="bf16x6", wgrad=None, bwd16=None):
        w = wgrad if wgrad is not None else conv
        if bwd16 is None:
            bwd16 = _BWD_F16 or conv == "f16x3"
        # (the weight-gradient kernels know fp32 and bf16x6; f16x3 reaches them through `bwd16`)
        return cls(_PREC[conv], BF16X6 if w == "f16x3" else _PREC[w], None, bool(bwd16) and conv != "fp32" and w != "fp32")

    @property
    def names(self):
        inv = {v: k for k, v in _PREC.items()}
        return inv[self.conv], ("f16x3" if self.bwd16 else inv[self.wgrad])


# the default is read ONCE from the environment and never mutated afterwards
DEFAULT_ARITH = Arith.of(os.environ.get("AZ_CONV_PRECISION", "f16x3"), os.environ.get("AZ_WGRAD_PRECISION"))


def _arith(a):
    if a is None:
        return DEFAULT_ARITH
    if isinstance(a, str):
        return Arith.of(a)
    return a


def _touched(*tensors):
    """Bump the version counters of buffers a kernel has just written through their raw pointers (BatchNorm
    running statistics and call counter): the memoised inference operands are keyed on them."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _dims(x):
    b, d, h, w, c = x.shape
    return b, d, h, w, c


def _out_dims(mode, d, h, w):
    if mode == CONV_S1:
        return d, h, w
    if mode == CONV_S2:
        return (d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return 2 * d, 2 * h, 2 * w


# Inference-time memo of packed weights and folded BatchNorm affine maps (pure functions of parameters
# that do not change between no_grad forwards).  Whether a call may use it is decided by the CALLER
# (`cache=` = "autograd is off at the call site", conv_bn / conv_plain), never inside an autograd
# Function.  Keys hold the tensors' storage address AND version counter, so an optimizer step or a
# load_state_dict (both write in place) invalidates them; a BatchNorm's key also holds
# num_batches_tracked, because this library's own train kernels update running_mean / running_var
# through raw pointers, which no version counter sees.  Each entry keeps its source tensors alive, so
# their addresses cannot be recycled for other data.  A lock makes the dicts safe under the threads of
# nn.DataParallel (train.py:540-541).
_PACK_CACHE, _AFFINE_CACHE = {}, {}
_CACHE_LOCK = threading.Lock()


def _cache_key(*tensors):
    return tuple((t.data_ptr(), t._version, t.device.index) for t in tensors if t is not None)


def _cache_get(cache, key):
    with _CACHE_LOCK:
        return cache.get(key)


def _cache_put(cache, key, value, limit):
    with _CACHE_LOCK:
        if len(cache) > limit:
            cache.clear()
        cache[key] = value


def _pack(weight, op_cin, op_cout, stride_out, stride_in, flip, precision, cache=False):
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if cache:
        key = (_cache_key(weight), op_cin, op_cout, stride_out, stride_in, bool(flip), precision)
        hit = _cache_get(_PACK_CACHE, key)
        if hit is not None:
            return hit[0]
    n = _lib.lib().az_conv3d_packed_floats(op_cin, op_cout, precision)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _call("az_conv3d_pack_weights", _p(packed), _p(w), op_cin, op_cout, stride_out, stride_in,
          int(flip), precision, _stream())
    packed.az_precision = precision  # the layout travels with the buffer: _run_gather launches what was packed
    if key is not None:
        _cache_put(_PACK_CACHE, key, (packed, weight), 256)
    return packed


def eval_affine(bn, like, cache=False):
    """(scale, shift) of an eval-mode BatchNorm: gamma/sqrt(running_var+eps), beta - running_mean*scale."""
    c = bn.num_features
    key = None
    if cache:
        key = (_cache_key(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked),
               float(bn.eps))
        hit = _cache_get(_AFFINE_CACHE, key)
        if hit is not None:
            return hit[0], hit[1]
    scale, shift = like.new_empty(c), like.new_empty(c)
    _call("az_bn3d_eval_affine", _p(scale), _p(shift), _p(bn.weight.detach()), _p(bn.bias.detach()),
          _p(bn.running_mean), _p(bn.running_var), float(bn.eps), c, _stream())
    if key is not None:
        _cache_put(_AFFINE_CACHE, key, (scale, shift, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        bn.num_batches_tracked), 512)
    return scale, shift


def _pack_forward(weight, mode, precision, cache=False, lazy=False, fits32=True):
    if mode == DECONV_S2:  # [Cin, Cout, 27]
        cin, cout = weight.shape[0], weight.shape[1]
        return _pack(weight, cin, cout, 27, cout * 27, False, precision, cache), cin, cout
    cout, cin = weight.shape[0], weight.shape[1]
    if not lazy:  # (the fused cost-volume operand, src = 1, stays on the 32x32x16 kernels)
        precision = _layout(precision, mode, cout, fits32)
    return _pack(weight, cin, cout, cin * 27, 27, False, precision, cache), cin, cout


def _peak(precision):
    """MFMA roofline of the arithmetic in use, in fp32-equivalent TFLOP/s: the fp32 MFMA dense
    peak, or the bf16 dense peak divided by the six bf16 MFMAs one fp32 product costs."""
    if precision == F16X3:
        return _PEAK_F16
    return (157.3, "fp32 MFMA") if precision == FP32 else (2500.0 / 6.0, "bf16x6: bf16 MFMA peak / 6")


def _conv_flops(b, vox_out, cin, cout, mode):
    taps = 27.0 / 8.0 if mode == DECONV_S2 else 27.0
    return 2.0 * taps * cin * cout * b * vox_out


class LazyCostVolume:
    """The PSMNet concat cost volume as a recipe (left/right NHWC features + number of
    disparity planes) instead of a [B,d,h,w,64] tensor: the first convolution
    synthesises its operand on the fly (az_conv3d_fwd, src = 1)."""

    def __init__(self, feat_l_nhwc, feat_r_nhwc, ndisp):
        self.fl, self.fr, self.ndisp = _chk(feat_l_nhwc, "feat_l"), _chk(feat_r_nhwc, "feat_r"), int(ndisp)
        if self.fl.shape != self.fr.shape or self.fl.shape[-1] != 32:
            raise RuntimeError("fused cost volume expects two [B,h,w,32] feature maps")


def _run_gather(x, packed, mode, cin, cout, precision, scale=None, shift=None, residual=None, relu=False,
                stats=False, tag="conv3d"):
    if isinstance(x, LazyCostVolume):
        if stats or mode != CONV_S1 or cin != 64:
            raise RuntimeError("the fused cost-volume operand is only wired for the eval-mode dres0[0] conv")
        b, h, w, _ = x.fl.shape
        d = x.ndisp
        out = x.fl.new_empty(b, d, h, w, cout)
        with profiler.scope(f"{tag}_costvol_m0_{cin}_{cout}", flops=_conv_flops(b, d * h * w, cin, cout, mode),
                            peak=_peak(precision)):
            _call("az_conv3d_fwd", _p(out), _p(x.fl), _p(x.fr), _p(packed), _p(scale), _p(shift),
                  _p(residual), int(relu), mode, 1, getattr(packed, "az_precision", precision), b, cin, cout, d, h, w,
                  _stream())
        return out
    b, d, h, w, c = _dims(x)
    assert c == cin, (c, cin)
    precision = getattr(packed, "az_precision", precision)  # (BF16X6_R16 for the layers _layout() routes there)
    if mode == CONV_S2 and ((d % 2 and d != 1) or h % 2 or w % 2):
        raise RuntimeError("stride-2 layers need even D/H/W (as PSMNet's hourglass does; D = 1: a 2-D layer)")
    do, ho, wo = _out_dims(mode, d, h, w)
    out = x.new_empty(b, do, ho, wo, cout)
    flops = _conv_flops(b, do * ho * wo, cin, cout, mode)
    name = f"{tag}_m{mode}_{cin}_{cout}"
    if stats:
        ntiles = _lib.lib().az_conv3d_stats_tiles(mode, precision, b, cin, cout, d, h, w)
        if ntiles <= 0:
            raise RuntimeError(f"az_conv3d_stats_tiles: {ntiles}")
        part = x.new_empty(cout, ntiles, 2)
        cnt = x.new_empty(ntiles)
        with profiler.scope(name, flops=flops, peak=_peak(precision)):
            _call("az_conv3d_fwd_stats", _p(out), _p(part), _p(cnt), _p(x), None, _p(packed), mode, 0,
                  precision, b, cin, cout, d, h, w, _stream())
        return out, part, cnt, ntiles
    with profiler.scope(name, flops=flops, peak=_peak(precision)):
        _call("az_conv3d_fwd", _p(out), _p(x), None, _p(packed), _p(scale), _p(shift), _p(residual),
              int(relu), mode, 0, precision, b, cin, cout, d, h, w, _stream())
    return out


# ---- f16x3 (include/azhip.h, az_roll_common.h) ---------------------------------------------------------------------
_PEAK_F16 = (2500.0 / 3.0, "f16x3: fp16 MFMA peak / 3")


AMAX_SLOTS = 1024  # include/azhip.h AZ_AMAX_FLOATS: floats of an "amax array" (16 slots, 256 bytes apart)


class _ZeroPool:
    """Pre-zeroed amax arrays for the kernels that take max |.| of what they write with atomicMax (BatchNorm apply,
    residual sums): one fill kernel per 256 arrays instead of a memset in front of every launch (measured:
    +0.1-0.16 ms per BatchNorm apply with the memset in the stream).  A slot is handed out once; the block lives as long
    as any of its slots."""

    def __init__(self):
        self.lock = threading.Lock()
        self.blocks = {}  # device index -> [tensor, next]

    def take(self, like):
        idx = like.device.index
        with self.lock:
            blk = self.blocks.get(idx)
            if blk is None or blk[1] + AMAX_SLOTS > blk[0].numel():
                blk = self.blocks[idx] = [torch.zeros(256 * AMAX_SLOTS, dtype=torch.float32, device=like.device), 0]
            i = blk[1]
            blk[1] = i + AMAX_SLOTS
            return blk[0][i:i + AMAX_SLOTS]


_ZEROS = _ZeroPool()
_W_AMAX = {}  # (data_ptr, version, device) -> device scalar max |w| of a weight tensor (lock: _CACHE_LOCK)


def _set_amax(t, am):
    """attach the device scalar max |t| to the tensor object (valid while the tensor's version counter stands)"""
    t.az_amax = (am, t._version)


def _get_amax(t):
    a = getattr(t, "az_amax", None)
    return a[0] if (a is not None and a[1] == t._version) else None


# AZ_DEBUG_AMAX=1 (read once): every amax attached to a tensor is checked against a fresh pass of az_absmax when it is
# used (synchronises; include/azhip.h, "CONTRACT of a caller-supplied amax": too small = clipped operands)
_DEBUG_AMAX = os.environ.get("AZ_DEBUG_AMAX", "0") not in ("", "0")


def check_amax(t, am):
    """raise if the amax array `am` is below the largest finite magnitude of t (a stale attribute)"""
    fresh = t.new_empty(AMAX_SLOTS)
    _call("az_absmax", _p(fresh), _p(t), t.numel(), _stream())
    have, true = float(am[::64].max()), float(fresh[::64].max())  # (the slots: every 64th float of an amax array)
    if not have >= true:
        raise RuntimeError(f"stale amax: {have:.6g} attached to a tensor whose largest finite magnitude is {true:.6g} "
                           "(written through a raw pointer or .data without conv3d._touched / a fresh conv3d.absmax?)")


def absmax(t):
    """device scalar max |t| (largest finite magnitude), the operand scale of the f16x3 kernels: taken by the kernel that
    produced t where that is one of this library's (BatchNorm apply / backward, residual sums), by a pass of az_absmax
    otherwise"""
    am = _get_amax(t)
    if am is not None and _DEBUG_AMAX:
        check_amax(t, am)
    if am is None:
        am = t.new_empty(AMAX_SLOTS)
        with profiler.scope("absmax", bytes=4.0 * t.numel(), bound="hbm"):
            _call("az_absmax", _p(am), _p(t), t.numel(), _stream())
        _set_amax(t, am)
    return am


# ---- pack plan (round 5): every f16x3 weight image of a model in ONE launch per optimizer step ----------------------------
# A training step used to pack each convolution weight twice (forward image, flipped / swapped input-gradient image), one
# launch and one allocation each: ~175 launches of ~4 us per step on the main stream.  The plan remembers, per device, which
# images the registered parameters were asked for (recorded by the first step's per-call packs), keeps ONE persistent buffer
# per image and a descriptor table on the device, and `prepack` rewrites all of them with az_pack_f16_multi when the
# parameters' version counters have moved.  _pack_f16 (here and in conv2d.py) then returns the plan's buffer.
#   * persistent buffers are safe in stream order: only main-stream kernels (forward, input gradient) read packed images,
#     and the next step's prepack is enqueued behind them on the same stream;
#   * only tensors registered through prepack() -- live nn.Parameters, held by weak reference -- have entries: derived
#     weights (the merged kernels of K3', DataParallel replicas) change address every step and keep the per-call route;
#   * an entry is valid for one (address, version counter): optimizer steps and load_state_dict write in place and bump it.
PACK_2D_SAME, PACK_2D_ROLL, PACK_3D_GATHER, PACK_3D_ROLL, PACK_3D_ROLL2 = 0, 1, 2, 3, 4
_PLAN_ON = os.environ.get("AZ_PACK_PLAN", "1") != "0"  # (read once) 0: every image packed by its own launch, as in round 4


class _PackEntry:
    __slots__ = ("wref", "kind", "cin", "cout", "ci_real", "co_real", "s_co", "s_ci", "taps", "flip", "packed", "amax",
                 "version", "index")


class PackPlan:
    """the images of one device"""
    DESC_BYTES = 72  # sizeof(AzPackDesc): 3 pointers, 2 long long, 8 ints

    def __init__(self, device):
        self.device = device
        self.lock = threading.RLock()
        self.registered = {}   # data_ptr -> weakref of the parameter
        self.entries = {}      # (data_ptr, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip) -> _PackEntry
        self.amax_rows = {}    # data_ptr -> (row tensor [AMAX_SLOTS], weakref)
        self.amax_blocks = []  # [tensor [256, AMAX_SLOTS], next row]
        self.table = None      # (descs, block_desc, first_block, nd, nblocks, entry list) on the device: the last one used
        self.tables = {}       # tuple of registered addresses -> table + the amax pointers it was built with
        self.launches = 0      # az_pack_f16_multi launches so far (tests)

    # -- weight amax rows: persistent, so that the descriptor table stays valid across steps
    def amax_row(self, w):
        ptr = w.data_ptr()
        hit = self.amax_rows.get(ptr)
        if hit is not None and hit[1]() is not None:
            return hit[0]
        if not self.amax_blocks or self.amax_blocks[-1][1] >= self.amax_blocks[-1][0].shape[0]:
            self.amax_blocks.append([torch.zeros(256, AMAX_SLOTS, dtype=torch.float32, device=self.device), 0])
        blk = self.amax_blocks[-1]
        row = blk[0][blk[1]]
        blk[1] += 1
        return row

    def alive(self, ptr):
        r = self.registered.get(ptr)
        return r is not None and r() is not None

    def purge(self):
        dead = [p for p, r in self.registered.items() if r() is None]
        if dead:
            dead = set(dead)
            for p in dead:
                del self.registered[p]
                self.amax_rows.pop(p, None)
            self.entries = {k: e for k, e in self.entries.items() if k[0] not in dead}
            self.table = None
            self.tables.clear()

    def lookup(self, weight, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip):
        """(entry, fresh): the entry of this image of a registered live parameter (created on first request), and whether its
        buffer already holds the image of the parameter's current version; None for unregistered tensors"""
        ptr = weight.data_ptr()
        with self.lock:
            if not self.alive(ptr):
                return None, False
            key = (ptr, kind, cin, cout, ci_real, co_real, int(s_co), int(s_ci), taps, bool(flip))
            e = self.entries.get(key)
            if e is None:
                e = _PackEntry()
                e.wref, e.kind, e.cin, e.cout, e.ci_real, e.co_real = self.registered[ptr], kind, cin, cout, ci_real, co_real
                e.s_co, e.s_ci, e.taps, e.flip = int(s_co), int(s_ci), taps, bool(flip)
                e.packed = torch.empty(taps * cin * cout, dtype=torch.float32, device=self.device)  # two fp16 parts per weight
                e.amax, e.version = None, -1
                self.entries[key] = e
                self.table = None  # rebuilt at the next prepack
                self.tables.clear()
            return e, e.version == weight._version

    def _build_table(self, todo):
        import numpy as np
        nd = len(todo)
        raw = np.zeros(nd * self.DESC_BYTES, dtype=np.uint8)
        q = raw.view(np.int64).reshape(nd, self.DESC_BYTES // 8)
        ints = raw.view(np.int32).reshape(nd, self.DESC_BYTES // 4)
        block_desc, first = [], []
        nblocks = 0
        for i, e in enumerate(todo):
            w = e.wref()
            q[i, 0], q[i, 1], q[i, 2] = e.packed.data_ptr(), w.data_ptr(), e.amax.data_ptr()
            q[i, 3], q[i, 4] = e.s_co, e.s_ci
            ints[i, 10:17] = (e.kind, e.cin, e.cout, e.ci_real, e.co_real, e.taps, int(e.flip))
            nb = (2 * e.taps * e.cin * e.cout + 255) // 256
            first.append(nblocks)
            block_desc.append(np.full(nb, i, dtype=np.int32))
            nblocks += nb
        dev = self.device
        self.table = (torch.from_numpy(raw).to(dev), torch.from_numpy(np.concatenate(block_desc)).to(dev),
                      torch.tensor(first, dtype=torch.int32, device=dev), nd, nblocks, list(todo))

    def prepack(self, weights):
        """register `weights` (nn.Parameters) and bring every recorded image of THEIRS up to their current version in one
        launch (a table per set of weights: another live model's images are not this call's business); the amax arrays of
        all of them in three launches (as prime_weight_amax)"""
        with self.lock:
            self.purge()
            import weakref
            mine = []
            for w in weights:
                if w is None or not w.is_cuda or w.dtype != torch.float32 or w.device != self.device:
                    continue
                r = self.registered.get(w.data_ptr())
                if r is None or r() is not w:
                    self.registered[w.data_ptr()] = weakref.ref(w)
                mine.append(w.data_ptr())
            setkey = tuple(sorted(set(mine)))
            owned = set(setkey)
            todo = [e for k, e in self.entries.items() if k[0] in owned and e.wref() is not None]
            stale = [e for e in todo if e.version != e.wref()._version]
            if not stale:
                return
            with torch.no_grad(), torch.cuda.device(self.device):
                # amax rows of the weights behind the stale images (persistent rows: slot 0 rewritten in place)
                ws = {}
                for e in stale:
                    w = e.wref()
                    ws[w.data_ptr()] = w
                wl = list(ws.values())
                rows = []
                for w in wl:
                    row = self.amax_row(w)
                    self.amax_rows[w.data_ptr()] = (row, self.registered[w.data_ptr()])
                    rows.append(row)
                maxes = torch._foreach_norm([w.detach() for w in wl], float("inf"))
                torch._foreach_copy_([r[0:1] for r in rows], [m.reshape(1) for m in maxes])
                for w, row in zip(wl, rows):
                    _cache_put(_W_AMAX, (w.data_ptr(), w._version, w.device.index, w.numel()), (row, w), 512)
                for e in stale:
                    e.amax = self.amax_rows[e.wref().data_ptr()][0]
                todo = [e for e in todo if e.amax is not None]
                tab = self.tables.get(setkey)
                if tab is None or tab[5] != todo or any(e.amax.data_ptr() != p for e, p in zip(todo, tab[6])):
                    self._build_table(todo)
                    tab = self.table + ([e.amax.data_ptr() for e in todo],)
                    if len(self.tables) > 8:
                        self.tables.clear()
                    self.tables[setkey] = tab
                descs, block_desc, first, nd, nblocks, elist, _ = tab
                _call("az_pack_f16_multi", _p(descs), _p(block_desc), _p(first), nd, nblocks, _stream())
                self.launches += 1
                for e in elist:
                    e.version = e.wref()._version
                self.table = tab[:6]  # (the last table used: tests read it)


_PLANS = {}


def pack_plan(device):
    with _CACHE_LOCK:
        p = _PLANS.get(device)
        if p is None:
            p = _PLANS[device] = PackPlan(device)
        return p


def prepack(weights):
    """start of a training forward pass: all f16x3 images of these parameters in one launch (PackPlan)"""
    ws = [w for w in weights if w is not None and w.is_cuda]
    if _PLAN_ON and ws:
        pack_plan(ws[0].device).prepack(ws)


def _planned_pack(weight, w, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip, pack_now):
    """the plan's (buffer, amax) for this image of a registered parameter, packed by `pack_now(packed, w_amax)` -- the
    per-tensor launch -- when the plan has not brought it up to date (first step, or no prepack this step); None for tensors
    the plan does not own"""
    if not _PLAN_ON or not w.is_cuda:
        return None
    plan = pack_plan(w.device)
    e, fresh = plan.lookup(weight, kind, cin, cout, ci_real, co_real, s_co, s_ci, taps, flip)
    if e is None:
        return None
    if not fresh:
        with plan.lock:
            wkey = (weight.data_ptr(), weight._version, weight.device.index, weight.numel())
            hit = _cache_get(_W_AMAX, wkey)
            if hit is not None:
                w_amax = hit[0]
            else:
                w_amax = absmax(w)
                _cache_put(_W_AMAX, wkey, (w_amax, weight), 512)
            pack_now(e.packed, w_amax)
            e.amax, e.version = w_amax, weight._version
            # (this entry's amax pointer may have changed: prepack compares the pointers its table was built with)
    return e.packed, e.amax


def prime_weight_amax(weights):
    """The amax arrays of all given weight tensors in THREE launches (one multi-tensor max-norm, one stack, one scatter
    into zeroed arrays) instead of a memset + reduction per weight and step: fills the per-version cache that the f16x3
    packers read.  Weights whose current version is already cached are skipped; called at the start of a forward pass."""
    todo = []
    for w in weights:
        if w is None or not w.is_cuda or w.dtype != torch.float32:
            continue
        key = (w.data_ptr(), w._version, w.device.index, w.numel())
        if _cache_get(_W_AMAX, key) is None:
            todo.append((key, w))
    if not todo:
        return
    with torch.no_grad():
        dets = [w.detach() for _, w in todo]
        maxes = torch.stack(torch._foreach_norm(dets, float("inf")))
        arr = torch.zeros(len(todo), AMAX_SLOTS, dtype=torch.float32, device=dets[0].device)
        arr[:, 0] = maxes
    for i, (key, w) in enumerate(todo):
        _cache_put(_W_AMAX, key, (arr[i], w), 512)


def _pack_f16(weight, op_cin, op_cout, stride_out, stride_in, flip, mode, cache=False):
    """`mode`: the index map the packed buffer will be launched with (it selects the kernel, hence the layout)"""
    w = _chk(weight.detach().contiguous(), "weight")
    key = None
    if cache:
        key = (_cache_key(weight), op_cin, op_cout, stride_out, stride_in, bool(flip), F16X3, mode)
        hit = _cache_get(_PACK_CACHE, key)
        if hit is not None:
            return hit[0]
    if not cache:
        kind = _lib.lib().az_conv3d_f16_layout(mode, op_cin, op_cout)
        if kind < 0:
            raise RuntimeError(f"az_conv3d_f16_layout({mode}, {op_cin}, {op_cout}): {kind}")
        hit = _planned_pack(weight, w, kind, op_cin, op_cout, op_cin, op_cout,
                            stride_out, stride_in, 27, flip,
                            lambda packed, w_amax: _call("az_conv3d_pack_weights_f16", _p(packed), _p(w), _p(w_amax), op_cin,
                                                         op_cout, stride_out, stride_in, int(flip), mode, _stream()))
        if hit is not None:
            return hit
    wkey = (weight.data_ptr(), weight._version, weight.device.index, weight.numel())
    hit = _cache_get(_W_AMAX, wkey)
    if hit is not None:
        w_amax = hit[0]
    else:  # once per weight and optimizer step: the forward launch computes it, the input gradient reuses it
        w_amax = absmax(w)
        _cache_put(_W_AMAX, wkey, (w_amax, weight), 512)
    packed = torch.empty(_lib.lib().az_conv3d_packed_floats_f16(op_cin, op_cout), dtype=torch.float32, device=w.device)
    _call("az_conv3d_pack_weights_f16", _p(packed), _p(w), _p(w_amax), op_cin, op_cout, stride_out, stride_in,
          int(flip), mode, _stream())
    if key is not None:
        _cache_put(_PACK_CACHE, key, ((packed, w_amax), weight), 256)
    return packed, w_amax


def _f16_fwd_ok(mode, cin, cout):
    """shapes with an f16x3 forward kernel (operand `cin` channels -> `cout`)"""
    return cin in (32, 64) and cout in (32, 64)


def _f16_dgrad_ok(mode, cin, cout):
    """layers whose input gradient has an f16x3 kernel: it is a forward launch dy [cout] -> dx [cin] of the dual map"""
    dual = {CONV_S1: CONV_S1, CONV_S2: DECONV_S2, DECONV_S2: CONV_S2}[mode]
    return _f16_fwd_ok(dual, cout, cin)


def _is_split(t):
    """the tensor holds pre-split fp16 pairs, not floats (set by _ConvBN.backward on the d(raw) it asked az_bn3d_bwd to
    write that way; the amax attached to it is the producer's bound)"""
    return getattr(t, "az_split", False)


def _run_f16(x, packed, w_amax, mode, cin, cout, scale=None, shift=None, residual=None, relu=False, stats=False,
             tag="conv3d"):
    b, d, h, w, c = _dims(x)
    assert c == cin, (c, cin)
    do, ho, wo = _out_dims(mode, d, h, w)
    out = x.new_empty(b, do, ho, wo, cout)
    name = f"{tag}_m{mode}_{cin}_{cout}"
    flops = _conv_flops(b, do * ho * wo, cin, cout, mode)
    split = _is_split(x)
    x_amax = _get_amax(x) if split else absmax(x)
    if split and (stats or x_amax is None):
        raise RuntimeError("a pre-split tensor reached a launch that cannot read it")
    if stats:
        ntiles = _lib.lib().az_conv3d_stats_tiles_f16(mode, b, cin, cout, d, h, w)
        if ntiles <= 0:
            raise RuntimeError(f"az_conv3d_stats_tiles_f16: {ntiles}")
        part, cnt = x.new_empty(cout, ntiles, 2), x.new_empty(ntiles)
        with profiler.scope(name, flops=flops, peak=_PEAK_F16):
            _call("az_conv3d_fwd_stats_f16", _p(out), _p(part), _p(cnt), _p(x), _p(packed), _p(x_amax), _p(w_amax),
                  mode, b, cin, cout, d, h, w, _stream())
        return out, part, cnt, ntiles
    with profiler.scope(name, flops=flops, peak=_PEAK_F16):
        _call("az_conv3d_fwd_f16", _p(out), _p(x), _p(packed), _p(x_amax), _p(w_amax), int(split), _p(scale), _p(shift),
              _p(residual), int(relu), mode, b, cin, cout, d, h, w, _stream())
    return out


def _conv(x, weight, mode, precision, scale=None, shift=None, residual=None, relu=False, stats=False, cache=False,
          tag="conv3d"):
    """forward convolution of a layer from its weight in PyTorch's layout, on `precision` where a kernel for the shape
    exists (f16x3 falls back to bf16x6 otherwise); returns what _run_gather returns"""
    lazy = isinstance(x, LazyCostVolume)
    if mode == DECONV_S2:
        cin, cout = weight.shape[0], weight.shape[1]
    else:
        cout, cin = weight.shape[0], weight.shape[1]
    fits = _fits32(x, cin, cout)
    if precision == F16X3 and not fits and mode == CONV_S1 and (cout == 32 or cin == cout == 64):
        precision = BF16X6  # (a batch element >= 4 GiB: the flat-address bf16x6 kernels)
    if precision == F16X3 and mode == DECONV_S2 and not lazy and not _fits32_transposed(x, cin, cout):
        precision = BF16X6
    if precision == F16X3 and not fits and _on_s2roll(CONV_S2 if mode == CONV_S2 else None, cin, cout):
        precision = BF16X6
    if precision == F16X3 and not lazy and _f16_fwd_ok(mode, cin, cout):
        if mode == DECONV_S2:
            pk, w_amax = _pack_f16(weight, cin, cout, 27, cout * 27, False, mode, cache)
        else:
            pk, w_amax = _pack_f16(weight, cin, cout, cin * 27, 27, False, mode, cache)
        return _run_f16(x, pk, w_amax, mode, cin, cout, scale, shift, residual, relu, stats, tag)
    prec = BF16X6 if precision == F16X3 else precision
    packed, cin, cout = _pack_forward(weight, mode, prec, cache=cache, lazy=lazy, fits32=fits)
    return _run_gather(x, packed, mode, cin, cout, prec, scale, shift, residual, relu, stats, tag)


def _input_grad_f16(dy, weight, mode, cin, cout, residual=None):
    """f16x3 form of _input_grad (shapes of _f16_dgrad_ok)"""
    if mode == CONV_S1:    # flipped taps, channels swapped
        pk, w_amax = _pack_f16(weight, cout, cin, 27, cin * 27, True, CONV_S1)
        return _run_f16(dy, pk, w_amax, CONV_S1, cout, cin, residual=residual, tag="dgrad")
    if mode == CONV_S2:    # transposed conv of dy with W[co][ci][k]
        pk, w_amax = _pack_f16(weight, cout, cin, 27, cin * 27, False, DECONV_S2)
        return _run_f16(dy, pk, w_amax, DECONV_S2, cout, cin, residual=residual, tag="dgrad")
    pk, w_amax = _pack_f16(weight, cout, cin, cout * 27, 27, False, CONV_S2)  # stride-2 conv of dy with Wt[ci][co][k]
    return _run_f16(dy, pk, w_amax, CONV_S2, cout, cin, residual=residual, tag="dgrad")


def _f16_wgrad_ok(mode, cin, cout):
    return cin in (32, 64) and cout in (32, 64)


def _wgrad_f16(coarse, fine, stride, cm, cn, tag, sink=None, late_ok=True):
    """late_ok = False: the caller reads the result on the side stream right away (a slice / permute of it is the real
    gradient): no deferred epilogue"""
    b, dc, hc, wc, _ = _dims(coarse)
    _, df, hf, wf, _ = _dims(fine)
    gw = coarse.new_empty(cm, cn, 3, 3, 3)
    ws_bytes = _lib.lib().az_conv3d_wgrad_workspace(cm, cn)
    am_c, am_f = _get_amax(coarse), _get_amax(fine)
    mask = (1 if _is_split(coarse) else 0) | (2 if _is_split(fine) else 0)
    if (mask & 1 and am_c is None) or (mask & 2 and am_f is None):
        raise RuntimeError("a pre-split tensor without its producer's amax")
    with overlap.scope(sink, coarse, fine, gw, am_c, am_f):
        # on the side stream the epilogue is deferred: zeroed workspace from the pass's arena, unpack at the join (overlap.Sink)
        late = late_ok and sink is not None and sink.live and overlap.DEFER_UNPACK
        ws = sink.take_workspace(ws_bytes // 4) if late else coarse.new_empty(ws_bytes // 4)
        # (a pass over the tensor only when its producer attached none)
        am_c = am_c if mask & 1 else absmax(coarse)
        am_f = am_f if mask & 2 else absmax(fine)
        with profiler.scope(f"{tag}_wgrad_s{stride}_{cm}_{cn}", flops=2.0 * 27 * cm * cn * b * dc * hc * wc,
                            peak=_PEAK_F16):
            _call("az_conv3d_wgrad_f16", None if late else _p(gw), _p(ws), ws_bytes, _p(coarse), _p(fine), _p(am_c), _p(am_f),
                  mask, stride, b, cm, cn, dc, hc, wc, df, hf, wf, _stream())
        if late:
            sink.defer_unpack(gw, ws, cm, cn, cm, cn, 27)
        if sink is not None and sink.live:
            sink.keep.extend((am_c, am_f))
    return gw


def _weight_grad_f16(x, dy, mode, cin, cout, sink=None, late_ok=True):
    if mode == DECONV_S2:
        return _wgrad_f16(x, dy, 2, cin, cout, "deconv", sink, late_ok)
    return _wgrad_f16(dy, x, 1 if mode == CONV_S1 else 2, cout, cin, "conv", sink, late_ok)


def _wgrad(coarse, fine, stride, cm, cn, tag, precision, sink=None):
    b, dc, hc, wc, _ = _dims(coarse)
    _, df, hf, wf, _ = _dims(fine)
    gw = coarse.new_empty(cm, cn, 3, 3, 3)
    ws_bytes = _lib.lib().az_conv3d_wgrad_workspace(cm, cn)
    with overlap.scope(sink, coarse, fine, gw):  # (gw too: the engine may drop it before the join, overlap.py)
        ws = coarse.new_empty(ws_bytes // 4)
        with profiler.scope(f"{tag}_wgrad_s{stride}_{cm}_{cn}", flops=2.0 * 27 * cm * cn * b * dc * hc * wc,
                            peak=_peak(precision)):
            _call("az_conv3d_wgrad", _p(gw), _p(ws), ws_bytes, _p(coarse), _p(fine), stride, precision,
                  b, cm, cn, dc, hc, wc, df, hf, wf, _stream())
    return gw


def _input_grad(dy, weight, mode, cin, cout, precision, residual=None):
    """gradient of the layer's input from the gradient dy of its (raw) convolution output; `residual` (a tensor of the
    result's shape) is added in the kernel's epilogue: another consumer's gradient of the same tensor (GradSlot)"""
    fits = _fits32(dy, cin, cout)
    if precision == F16X3:
        if _f16_dgrad_ok(mode, cin, cout) and (fits or not (mode == CONV_S1 and (cin == 32 or cin == cout == 64))) and \
                (mode != CONV_S2 or _fits32_transposed(dy, cout, cin)) and \
                (fits or not _on_s2roll(CONV_S2 if mode == DECONV_S2 else None, cout, cin)):
            return _input_grad_f16(dy, weight, mode, cin, cout, residual)
        precision = BF16X6
    if mode == CONV_S1:    # flipped taps, channels swapped
        pk = _pack(weight, cout, cin, 27, cin * 27, True, _layout(precision, CONV_S1, cin, fits))
        return _run_gather(dy, pk, CONV_S1, cout, cin, precision, residual=residual, tag="dgrad")
    if mode == CONV_S2:    # transposed conv of dy with W[co][ci][k]
        pk = _pack(weight, cout, cin, 27, cin * 27, False, precision)
        return _run_gather(dy, pk, DECONV_S2, cout, cin, precision, residual=residual, tag="dgrad")
    pk = _pack(weight, cout, cin, cout * 27, 27, False, precision)  # stride-2 conv of dy with Wt[ci][co][k]
    return _run_gather(dy, pk, CONV_S2, cout, cin, precision, residual=residual, tag="dgrad")


# ---- gradient hand-over between the consumers of one tensor (round 5) -----------------------------------------------------
# A tensor with several consumers gets one gradient per consumer and the autograd engine adds them pairwise: 8 `add_` kernels
# over V0- / V1-sized tensors per step (1.8 ms on the main stream, tools/aten_sources.py).  Every input-gradient kernel can add
# a tensor in its epilogue, so the consumers of this module do the sum themselves: each one that is NOT the last to run keeps
# its contribution in the tensor's GradSlot and returns None to the engine; the last one launches its input gradient with the
# kept contribution(s) as `residual` and returns the total.  Correct for any execution order; a contribution that is a plain
# tensor (the residual branch of a BatchNorm unit) is handed over the same way.  What the engine sees is a sum whose other terms
# are None.  Safety net for pruned graphs (a consumer that registered in forward and never runs in backward): the producing
# _ConvBN node adds whatever is still parked in its output's slot to the gradient it receives.
_HANDOVER = os.environ.get("AZ_GRAD_HANDOVER", "1") != "0"  # (read once) 0: every consumer returns its own gradient


class GradSlot:
    __slots__ = ("expect", "left", "parts")

    def __init__(self):
        self.expect, self.left, self.parts = 0, 0, []


def _slot_register(t):
    """forward: one more consumer of t whose backward will call _slot_contribute"""
    if not (_HANDOVER and torch.is_grad_enabled() and t.requires_grad):
        return None
    s = getattr(t, "az_gslot", None)
    if s is None:
        node = t.grad_fn
        # only tensors produced by a _ConvBN node: that node is the safety net (_slot_leftover) for contributions parked by
        # consumers whose siblings the engine pruned; everything else keeps the engine's own accumulation
        if node is None or type(node).__name__ != "_ConvBNBackward":
            return None
        s = t.az_gslot = GradSlot()
        node.az_out_slot = s
    s.expect += 1
    s.left = s.expect
    return s


def _sum_parts(parts):
    while len(parts) > 1:
        take, parts = parts[:4], parts[4:]
        out = torch.empty_like(take[0])
        ptrs = [_p(_chk(t.contiguous(), "grad")) for t in take] + [None] * (4 - len(take))
        _call("az_sum4", _p(out), ptrs[0], ptrs[1], ptrs[2], ptrs[3], out.numel(), _stream())
        parts = [out] + parts
    return parts[0]


def _slot_contribute(slot, make=None, plain=None):
    """backward of one consumer: `make(residual)` launches its input gradient with `residual` added in the epilogue, or `plain`
    is a gradient that already exists.  Returns what this consumer hands to the engine (None unless it is the last one)."""
    if slot is None or slot.expect < 2:
        return make(None) if make is not None else plain
    slot.left -= 1
    if slot.left > 0:
        slot.parts.append(make(None) if make is not None else plain)
        return None
    parts, slot.parts, slot.left = slot.parts, [], slot.expect
    res = _sum_parts(parts) if parts else None
    if make is not None:
        return make(res)
    return plain if res is None else _sum_parts([plain, res])


def _slot_leftover(ctx, gy):
    """producer side: contributions parked in the output's slot by consumers whose siblings never ran (pruned graphs)"""
    slot = getattr(ctx, "az_out_slot", None)
    if slot is None or not slot.parts:
        return gy
    parts, slot.parts, slot.left = slot.parts, [], slot.expect
    return _sum_parts([gy] + parts)


def _weight_grad(x, dy, mode, cin, cout, precision, sink=None, late_ok=True):
    if precision == F16X3:
        if _f16_wgrad_ok(mode, cin, cout):
            return _weight_grad_f16(x, dy, mode, cin, cout, sink, late_ok)
        precision = BF16X6
    if mode == DECONV_S2:
        return _wgrad(x, dy, 2, cin, cout, "deconv", precision, sink)
    return _wgrad(dy, x, 1 if mode == CONV_S1 else 2, cout, cin, "conv", precision, sink)


def _presplit_ok(x, raw, mode, cin, cout, need_gx, need_gw):
    """d(raw) of this layer may be written pre-split: every kernel that will read it -- the f16x3 input gradient (a forward
    launch of the dual map on d(raw)) and the f16x3 weight gradient -- stages such an operand by copy.  Mirrors the
    routing of _input_grad / _weight_grad; the library answers for its own kernels (az_conv3d_*_split_ok)."""
    lib = _lib.lib()
    if lib.az_option(b"AZ_BN_BWD_FUSED") == 0:  # (the three-launch BatchNorm backward of that A/B switch writes floats only)
        return False
    b, d, h, w, _ = raw.shape
    if need_gx:
        if not (_f16_dgrad_ok(mode, cin, cout) and (_fits32(raw, cin, cout) or not (mode == CONV_S1 and (cin == 32 or cin == cout == 64))) and
                (mode != CONV_S2 or _fits32_transposed(raw, cout, cin)) and
                (_fits32(raw, cin, cout) or not _on_s2roll(CONV_S2 if mode == DECONV_S2 else None, cout, cin))):
            return False
        dual = {CONV_S1: CONV_S1, CONV_S2: DECONV_S2, DECONV_S2: CONV_S2}[mode]
        if lib.az_conv3d_fwd_f16_split_ok(dual, b, cout, cin, d, h, w) != 1:
            return False
    if need_gw:
        if not _f16_wgrad_ok(mode, cin, cout):
            return False
        _, dx_, hx, wx, _ = x.shape
        if mode == DECONV_S2:   # coarse = x, fine = d(raw)
            ok, bit = lib.az_conv3d_wgrad_f16_split_ok(2, b, cin, cout, dx_, hx, wx, d, h, w), 2
        else:                   # coarse = d(raw), fine = x
            ok, bit = lib.az_conv3d_wgrad_f16_split_ok(1 if mode == CONV_S1 else 2, b, cout, cin, d, h, w, dx_, hx, wx), 1
        if not ok & bit:
            return False
    return bool(need_gx or need_gw)


class _ConvBN(torch.autograd.Function):
    """y = relu?( BN(conv(x, weight)) + residual ), one autograd node per convbn_3d unit."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, mode, relu, arith, defer=None, x_slot=None, res_slot=None):
        ctx.x_slot, ctx.res_slot = x_slot, res_slot  # (registered by conv_bn, where autograd is on: GradSlot)
        x = _chk(x, "x")  # (LazyCostVolume never reaches autograd: see conv_bn)
        if residual is not None:
            residual = _chk(residual, "residual")
        training = bn.training or not bn.track_running_stats
        eps = float(bn.eps)
        if mode == DECONV_S2:
            cin, cout = weight.shape[0], weight.shape[1]
        else:
            cout, cin = weight.shape[0], weight.shape[1]
        with torch.cuda.device(x.device):
            if not training:
                # eval-mode BatchNorm under autograd (fine-tuning with frozen statistics): the same
                # kernels, with the running-statistics affine map; backward below
                scale, shift = eval_affine(bn, x)
                raw = _conv(x, weight, mode, arith.conv)
                y = torch.empty_like(raw)
                y_amax = _ZEROS.take(raw)
                _call("az_bn3d_apply", _p(y), _p(raw), _p(scale), _p(shift), _p(residual), int(relu),
                      raw.numel() // cout, cout, _p(y_amax), _stream())
                _set_amax(y, y_amax)
                ctx.save_for_backward(x, weight, gamma, raw, y if relu else None, bn.running_mean.clone(),
                                      scale, None, None)
                ctx.cfg = (mode, relu, residual is not None, cin, cout, arith, False)
                return y
            raw, part, cnt, ntiles = _conv(x, weight, mode, arith.conv, stats=True)
            scale, shift = x.new_empty(cout), x.new_empty(cout)
            mean, invstd = x.new_empty(cout), x.new_empty(cout)
            track = bn.track_running_stats and bn.running_mean is not None
            momentum = 0.1 if bn.momentum is None else float(bn.momentum)
            nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
            scratch = x.new_empty(int(_lib.lib().az_bn3d_finalize_scratch(cout))) if ntiles >= 4096 else None
            _call("az_bn3d_finalize", _p(mean), _p(invstd), _p(scale), _p(shift),
                  _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                  _p(part), _p(cnt), _p(gamma.detach()), _p(beta.detach()), ntiles, cout, eps,
                  momentum, _p(nbt), _p(scratch), scratch.numel() if scratch is not None else 0, _stream())
            if nbt is not None:  # the kernel wrote through raw pointers: tell the version counters (cache keys)
                _touched(nbt, bn.running_mean, bn.running_var)
            nvox = raw.numel() // cout
            if defer is not None and relu and residual is None:
                # the single consumer applies relu(raw * scale + shift) itself while it stages its input
                # (DeferredAffine): no apply pass, no second copy of the tensor; this node's output IS raw
                defer.scale, defer.shift = scale, shift
                y = raw
            else:
                y = torch.empty_like(raw)
                y_amax = _ZEROS.take(raw)
                with profiler.scope(f"bn3d_apply_{cout}", bytes=4.0 * raw.numel() * (3 if residual is not None else 2),
                                    bound="hbm"):
                    _call("az_bn3d_apply", _p(y), _p(raw), _p(scale), _p(shift), _p(residual), int(relu),
                          nvox, cout, _p(y_amax), _stream())
                _set_amax(y, y_amax)
        # a ReLU layer without residual recomputes its mask from raw in backward (fma(raw, scale, shift) > 0)
        remask = relu and residual is None
        ctx.save_for_backward(x, weight, gamma, raw, y if (relu and not remask) else None, mean, invstd,
                              scale if remask else None, shift if remask else None)
        if arith.sink is not None and not arith.sink.owns(weight):  # see conv2d._leaf_sink
            arith = arith._replace(sink=None)
        ctx.cfg = (mode, relu, residual is not None, cin, cout, arith, True)
        # the f16x3 weight gradient scales its operand x by max |x|: keep what the forward pass knows of it
        ctx.x_amax = _get_amax(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, gamma, raw, y, mean, invstd, scale, shift = ctx.saved_tensors
        mode, relu, has_res, cin, cout, arith, training = ctx.cfg
        gy = _chk(gy.contiguous(), "grad_y")
        nvox = raw.numel() // cout
        with torch.cuda.device(gy.device):
            gy = _slot_leftover(ctx, gy)
            if not training:
                # y = relu?(raw * s + t + res) with s = gamma * rinv, t = beta - running_mean * s (constants
                # of the running statistics): dz = gy * [y > 0]; d raw = dz * s; d gamma = sum dz (raw - rm) rinv;
                # d beta = sum dz.  A rarely used path (frozen-BN fine-tuning): plain tensor ops.
                s_, rm = invstd, mean  # saved slots: (running_mean copy, scale)
                dz = gy * (y > 0).to(gy.dtype) if relu else gy
                dbeta = dz.sum(dim=(0, 1, 2, 3))
                rinv = torch.where(gamma.detach() != 0, s_ / gamma.detach(), torch.zeros_like(s_))
                dgamma = (dz * (raw - rm)).sum(dim=(0, 1, 2, 3)) * rinv
                dx_raw = (dz * s_).contiguous()
                g_res = dz if has_res else None
            else:
                lib = _lib.lib()
                dx_raw = torch.empty_like(raw)
                dz = torch.empty_like(raw) if (has_res and relu) else None
                dgamma, dbeta = gy.new_empty(cout), gy.new_empty(cout)
                coef = gy.new_empty(cout, 3)
                ws_bytes = lib.az_bn3d_bwd_workspace(nvox, cout)
                ws = gy.new_empty(ws_bytes // 4)
                # f16x3 gradients: max |dx_raw| (the operand scale) is taken by the kernel that writes dx_raw
                dx_amax = gy.new_empty(AMAX_SLOTS) if arith.bwd16 else None
                # ... and dx_raw itself is written pre-split where both of its readers stage such a tensor by copy
                split = arith.bwd16 and PRESPLIT and _presplit_ok(x, raw, mode, cin, cout, ctx.needs_input_grad[0],
                                                                  ctx.needs_input_grad[1])
                with profiler.scope(f"bn3d_bwd_{cout}", bytes=4.0 * raw.numel() * (7 if (relu and y is not None) else 5), bound="hbm"):
                    _call("az_bn3d_bwd", _p(dx_raw), _p(dz), _p(dgamma), _p(dbeta), _p(coef), _p(ws), ws_bytes,
                          _p(gy), _p(y), _p(raw), _p(mean), _p(invstd), _p(gamma.detach()), _p(scale), _p(shift),
                          int(relu), nvox, cout, _p(dx_amax), int(split), _stream())
                if dx_amax is not None:
                    _set_amax(dx_raw, dx_amax)
                if split:
                    dx_raw.az_split = True
                g_res = (dz if relu else gy) if has_res else None
            gx = gw = None
            # (the residual branch first: when this layer's own input is the same tensor's last consumer, it adds it)
            if has_res:
                g_res = _slot_contribute(ctx.res_slot, plain=g_res)
            if ctx.needs_input_grad[0]:
                gx = _slot_contribute(ctx.x_slot, make=lambda r: _input_grad(dx_raw, weight, mode, cin, cout,
                                                                             F16X3 if arith.bwd16 else arith.conv, r))
            if ctx.needs_input_grad[1]:
                if getattr(ctx, "x_amax", None) is not None and _get_amax(x) is None:
                    _set_amax(x, ctx.x_amax)
                gw = _weight_grad(x, dx_raw, mode, cin, cout, F16X3 if arith.bwd16 else arith.wgrad, arith.sink)
        return gx, gw, dgamma, dbeta, g_res, None, None, None, None, None, None, None


class DeferredAffine:
    """Filled by conv_bn(..., defer=this) in train mode: the returned tensor is the RAW convolution output and
    its one consumer must take relu(raw * scale + shift) (conv_logits(..., affine=this))."""

    def __init__(self):
        self.scale = self.shift = None

    @property
    def pending(self):
        return self.scale is not None


def conv_bn(x, conv, bn, mode, relu=False, residual=None, arith=None, defer=None):
    arith = _arith(arith)
    if isinstance(x, LazyCostVolume):  # inference: BN folded, operand synthesised in-kernel
        if torch.is_grad_enabled() or bn.training:
            raise RuntimeError("LazyCostVolume is an inference-only operand")
        with torch.cuda.device(x.fl.device):
            scale, shift = eval_affine(bn, x.fl, cache=True)
            return _conv(x, conv.weight, mode, arith.conv, scale, shift, residual, relu, cache=True)
    training = bn.training or not bn.track_running_stats
    if not training and not torch.is_grad_enabled():
        # inference: BatchNorm folded into the conv epilogue, packed weights / affine maps memoised
        x = _chk(x, "x")
        with torch.cuda.device(x.device):
            scale, shift = eval_affine(bn, x, cache=True)
            return _conv(x, conv.weight, mode, arith.conv, scale, shift,
                         _chk(residual, "residual") if residual is not None else None, relu, cache=True)
    w = arith.sink.weight(conv.weight) if arith.sink is not None else conv.weight
    x_slot = _slot_register(x)
    res_slot = _slot_register(residual) if residual is not None else None
    return _ConvBN.apply(x, w, bn.weight, bn.bias, residual, bn, mode, relu, arith, defer, x_slot, res_slot)


class _ConvLogits(torch.autograd.Function):
    """logits = Conv3d(32 -> 1)(x) + addend  (classifN[2] and the running cost sums)."""

    @staticmethod
    def forward(ctx, x, weight, addend, sink, scale=None, shift=None):
        ctx.sink = sink if (sink is not None and sink.owns(weight)) else None
        x = _chk(x, "x")
        w = _chk(weight.detach().contiguous(), "weight")
        b, d, h, wd, c = _dims(x)
        if c != 32 or tuple(w.shape) != (1, 32, 3, 3, 3):
            raise RuntimeError("classifier conv expects 32 -> 1 channels, 3x3x3")
        if addend is not None:
            addend = _chk(addend.contiguous(), "addend")
        out = x.new_empty(b, d, h, wd)
        with torch.cuda.device(x.device):
            with profiler.scope("conv3d_c1_fwd", bytes=4.0 * (x.numel() + out.numel()), bound="hbm"):
                _call("az_conv3d_c1_fwd", _p(out), _p(x), _p(w), _p(addend), _p(scale), _p(shift), b, d, h, wd, _stream())
        ctx.save_for_backward(x, w, scale, shift)
        ctx.has_add = addend is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, scale, shift = ctx.saved_tensors
        g = _chk(g.contiguous(), "grad_logits")
        b, d, h, wd, _ = _dims(x)
        gx = gw = None
        with torch.cuda.device(g.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                with profiler.scope("conv3d_c1_dgrad", bytes=4.0 * (x.numel() + g.numel()), bound="hbm"):
                    _call("az_conv3d_c1_dgrad", _p(gx), _p(g), _p(w), b, d, h, wd, _stream())
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(w)
                # (scale / shift too: 32-float saved tensors that the engine frees right after this node -- the main
                #  stream's next small allocation would overwrite them before the queued side-stream kernel reads them)
                with overlap.scope(ctx.sink, x, g, gw, scale, shift), \
                        profiler.scope("conv3d_c1_wgrad", bytes=4.0 * (x.numel() + g.numel()), bound="hbm"):
                    _call("az_conv3d_c1_wgrad", _p(gw), _p(x), _p(g), _p(scale), _p(shift), b, d, h, wd, _stream())
        return gx, gw, (g if ctx.has_add else None), None, None, None


def conv_logits(x, conv, addend=None, sink=None, affine=None):
    """affine: a DeferredAffine filled by the conv_bn that produced x (x is then its raw output)."""
    sc, sh = (affine.scale, affine.shift) if (affine is not None and affine.pending) else (None, None)
    return _ConvLogits.apply(x, sink.weight(conv.weight) if sink is not None else conv.weight, addend, sink, sc, sh)


class _AddRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "a"), _chk(b, "b")
        y = torch.empty_like(a)
        y_amax = _ZEROS.take(a)
        with torch.cuda.device(a.device):
            _call("az_add_relu", _p(y), _p(a), _p(b), 0, a.numel(), _p(y_amax), _stream())
        _set_amax(y, y_amax)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddRelu.apply(a, b)


class _FanOut(torch.autograd.Function):
    """n aliases of x whose gradients are summed in ONE kernel pass (az_sum4) instead of autograd's
    n-1 pairwise adds: for a V0 tensor with four consumers that is 5 tensor passes instead of 9."""

    @staticmethod
    def forward(ctx, x, n):
        views = tuple(x.view_as(x) for _ in range(n))
        am = _get_amax(x)
        if am is not None:
            for v in views:
                _set_amax(v, am)
        return views

    @staticmethod
    def backward(ctx, *grads):
        gs = [_chk(g.contiguous(), "grad") for g in grads if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            take, gs = gs[:4], gs[4:]
            out = torch.empty_like(take[0])
            ptrs = [_p(t) for t in take] + [None] * (4 - len(take))
            with torch.cuda.device(out.device):
                _call("az_sum4", _p(out), ptrs[0], ptrs[1], ptrs[2], ptrs[3], out.numel(), _stream())
            gs.insert(0, out)
        return gs[0], None


def fanout(x, n):
    """n views of x for n consumers (training); see _FanOut."""
    if not (torch.is_grad_enabled() and x.requires_grad) or x.numel() % 4:
        return (x,) * n
    return _FanOut.apply(x, n)


def conv_plain(x, weight, mode, arith=None):
    """Bare convolution (no BN), forward only -- used by parity tests and tools."""
    arith = _arith(arith)
    return _conv(_chk(x, "x"), weight, mode, arith.conv)
