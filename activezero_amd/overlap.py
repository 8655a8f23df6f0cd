"""Weight-gradient kernels on a second HIP stream, overlapped with the rest of the backward pass.

The backward pass of the path alternates MFMA-bound kernels (input gradients, weight gradients) with
HBM-bound ones (BatchNorm reductions and applies, gradient sums).  Only the chain
    input gradient of layer L -> BatchNorm backward of layer L-1 -> input gradient of layer L-1 -> ...
is sequentially dependent; the weight gradient of layer L has no consumer until the optimizer step.  With a
`Sink`, every weight-gradient kernel is launched on a side stream (after that stream has waited for the
main stream's current position, i.e. for its operands) and the main stream continues at once: matrix work
then runs beside the streaming kernels of the following layers (measured: 133.9 -> 126.5 ms per step at
B=4, 544x960, D=192).

Autograd and DistributedDataParallel see nothing unusual, by construction of the graph rather than by
convention:
  * `begin()` runs at the start of a forward pass.  It wraps every convolution weight w in a `_Gate` node (an
    identity whose output g(w) the layers use in place of w) and puts ONE `_Tail` node into the graph: all gated
    weights are its inputs, its single output (a 1-element token) is an input of the first convolution of the
    model.  In the backward pass the first convolution's node is necessarily the last convolution node to run
    (every other one consumes, directly or not, what it produced), so `_Tail.backward` runs after every
    weight-gradient kernel has been launched; there the main stream waits for the side stream.
  * Each `_Gate` therefore has two incoming gradient edges -- the real gradient from its layer's node and an empty
    one from `_Tail` -- and the engine runs `_Gate.backward` only after BOTH have arrived, i.e. after the join.
    Only then does the gradient travel on to the weight's AccumulateGrad (and DDP's hook behind it).  Between the
    layer's node and the gate nothing reads the tensor: the gate's input buffer holds a single defined gradient,
    so the engine has nothing to add up early -- also when several forward passes of the same module are
    back-propagated together (each pass has its own gates, tail and join).
  * Operands AND outputs of side-stream kernels are kept referenced by the pass's `Sink` until the join (the default,
    AZ_SIDE_RELEASE=join), so the caching allocator cannot hand their memory to later main-stream work while the side
    stream still reads or writes it (the outputs matter when the engine drops a weight gradient at once,
    `backward(inputs=[...])` on a subset: tests/test_gpu_overlap.py::test_partial_backward_still_joins_the_side_stream
    failed with 1e34-sized BatchNorm gradients before they were protected).  AZ_SIDE_RELEASE=record protects the
    OPERANDS with `Tensor.record_stream` instead, which frees each layer's operands as soon as the side stream has
    passed them (3.8 GB less at the allocated peak) but makes the allocator call hipMalloc inside the steps
    (why it is not the default); outputs are held until the join either way.
Backstops, so that the scheme cannot silently corrupt memory outside the case it was designed around:
  * the first kernel a pass sends to the side stream registers `sink.join` as an end-of-backward callback of the
    autograd engine: the join also happens when the engine never reaches `_Tail` (`backward(inputs=[...])` or
    `autograd.grad` on a subset that prunes the first convolution / the gates: the convolution nodes still launch
    their weight-gradient kernels because `needs_input_grad[1]` is set, but their results are dropped).  `join` is
    idempotent;
  * a gated weight that is requested twice in one pass (a module whose forward runs a convolution twice, e.g. the
    unequal-shape fallback of `forward_pair`) would give its gate two REAL gradient edges, which the engine sums
    as soon as the second arrives -- before the join.  The sink then disarms itself: every weight gradient of that
    pass stays in order;
  * `begin()` first makes the main stream wait for the side stream, so that a pass that died part-way through its
    backward (exception: no callback runs) cannot leak un-joined work into the next one.
There is no module-level state: a Sink belongs to one forward/backward pass of one module replica (the side
stream itself is cached per device).  `begin()` returns None -- plain in-order weight gradients -- whenever the
construction above does not apply (no grad mode, a frozen convolution weight, CPU tensors).
"""
import os
import threading

import torch

from . import profiler

_STREAMS = {}
_STREAMS_LOCK = threading.Lock()
# HIP priority of the side stream (read once): larger = lower.  The backward pass's critical chain is the MAIN stream
# (input gradients and BatchNorm backward); the weight gradients have slack until the join at the end.
_SIDE_PRIORITY = int(os.environ.get("AZ_SIDE_PRIORITY", "0"))
# How the operands of a side-stream kernel (allocated on, and returned to, the MAIN stream's allocator pool) are protected
# while that kernel has not run yet -- AZ_SIDE_RELEASE, read once:
#   "join"   (default) Python references until the join at the end of backward: the operands of every layer (x, dy: 0.2-0.8 GB
#            each at V0) stay allocated until then -- 32.9 GB at the peak of a B = 4 step -- and the allocator sees ordinary frees;
#   "record" Tensor.record_stream: each operand is released as soon as autograd drops it, 29.1 GB allocated at the peak -- but the
#            host runs several layers ahead of the GPU, the caching allocator holds such a block back until ITS event poll finds
#            the side stream past the free, requests in between miss the pool and it calls hipMalloc: 75 calls inside 8 timed
#            steps after 3 warm-up steps, 9 after 12 (profiles/r04_side_release_ab.txt).  Free on a fresh box; +50 ms per step
#            on a box whose memory a previous process has just released.  For memory-constrained runs only.
# (A third form -- references dropped when an event behind the kernel has completed, polled at every later launch -- behaves as
#  "join": at enqueue time the GPU is far behind, nothing has completed yet.)
_RELEASE = os.environ.get("AZ_SIDE_RELEASE", "join")
# AZ_WGRAD_DEFER=0 (read once): every side-stream weight gradient zeroes its own workspace and unpacks at once (round 4)
DEFER_UNPACK = os.environ.get("AZ_WGRAD_DEFER", "1") != "0"
if _RELEASE not in ("record", "join"):
    _RELEASE = "join"


def side_stream(device):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    with _STREAMS_LOCK:
        s = _STREAMS.get(idx)
        if s is None:
            s = _STREAMS[idx] = torch.cuda.Stream(device=idx, priority=_SIDE_PRIORITY)
    return s


_ARENA_HINT = {}   # device index -> floats of weight-gradient workspace the last pass used


class Sink:
    """State of one forward/backward pass: the side stream, the token that ties `_Tail` to the first
    convolution, and the operands kept alive until the join."""

    def __init__(self, device):
        self.stream = side_stream(device)
        self.token = None
        self.gated = {}      # id(weight) -> (weight, gated alias) of this pass
        self.owned = set()   # id(gated alias)
        self.keep = []
        self.armed = False   # set by the first convolution when it takes the token: without that edge in the
        self.joined = False  # graph nothing would ever join the side stream, so nothing is sent there
        self.uses = {}       # id(weight) -> times requested in this pass
        self.disarmed = False
        self.callback_set = False
        # round 5: one zeroed arena for all weight-gradient workspaces of the pass and one unpack launch at the join
        self.arena = None    # flat fp32 tensor, allocated and zeroed on the side stream by the first weight gradient
        self.arena_used = 0
        self.arena_need = 0  # floats this pass asked for (the next pass's arena size)
        self.pending = []    # (grad_w, workspace view, cm, cn, cm_real, cn_real, taps): unpacked at the join

    def weight(self, w):
        """The tensor a layer of this pass must use for parameter w."""
        hit = self.gated.get(id(w))
        if hit is None:
            return w
        n = self.uses.get(id(w), 0) + 1
        self.uses[id(w)] = n
        if n > 1:  # two gradient edges into one gate: the engine would add them before the join
            self.disarmed = True
        return hit[1]

    @property
    def live(self):
        """weight-gradient kernels of this pass may go to the side stream"""
        return self.armed and not self.joined and not self.disarmed

    def owns(self, t):
        """True when t is one of this pass's gated weights: only their gradients may be produced late."""
        return id(t) in self.owned

    # ---- deferred weight-gradient epilogues (call both INSIDE `scope`, i.e. with the side stream current) -------------------
    def take_workspace(self, nfloats):
        """a ZEROED tap-major workspace for a weight-gradient kernel launched with grad_w = NULL (include/azhip.h): a slice of
        the pass's arena -- one fill per backward pass instead of a memset per layer"""
        n = (nfloats + 63) & ~63
        self.arena_need += n
        if self.arena is None:
            dev = self.stream.device
            self.arena = torch.zeros(max(_ARENA_HINT.get(dev.index, 0), n), dtype=torch.float32, device=dev)
            self.arena_used = 0
        if self.arena_used + n > self.arena.numel():  # (first pass, or a pass larger than the last one)
            return torch.zeros(n, dtype=torch.float32, device=self.arena.device)
        ws = self.arena[self.arena_used:self.arena_used + n]
        self.arena_used += n
        return ws

    def defer_unpack(self, grad_w, ws, cm, cn, cm_real, cn_real, taps):
        """grad_w (PyTorch layout, still unwritten) <- ws at the join; both are kept alive until then"""
        self.pending.append((grad_w, ws, cm, cn, cm_real, cn_real, taps))

    def _flush_pending(self):
        import numpy as np
        from .ops import _call, _p
        pend, self.pending = self.pending, []
        nd = len(pend)
        raw = np.zeros(nd * 40, dtype=np.uint8)  # sizeof(AzUnpackDesc): 2 pointers, 6 ints
        q, ints = raw.view(np.int64).reshape(nd, 5), raw.view(np.int32).reshape(nd, 10)
        block_desc, first, nblocks = [], [], 0
        for i, (gw, ws, cm, cn, cmr, cnr, taps) in enumerate(pend):
            q[i, 0], q[i, 1] = gw.data_ptr(), ws.data_ptr()
            ints[i, 4:9] = (cm, cn, cmr, cnr, taps)
            nb = (cmr * cnr * taps + 255) // 256
            first.append(nblocks)
            block_desc.append(np.full(nb, i, dtype=np.int32))
            nblocks += nb
        tables = np.concatenate([raw.view(np.int32), np.concatenate(block_desc), np.asarray(first, dtype=np.int32)])
        dev = self.stream.device
        # a FRESH pinned tensor per join, from torch's caching host allocator: the asynchronous copy below executes when the side
        # stream reaches it -- tens of milliseconds after this line, and the host may be several steps ahead of the GPU by then
        # -- and that allocator does not recycle a block before the copies recorded on it have completed.  (A hand-kept pair of
        # staging buffers was overwritten by the join of step k + 2 before step k's copy had run: gradients unpacked into the
        # wrong tensors, invisible in tests that synchronise every step, 11.6 -> 14.3 in the bench's loss after 13 steps.)
        stage = torch.from_numpy(tables).pin_memory()
        with torch.cuda.stream(self.stream):
            t = stage.to(dev, non_blocking=True)
            o1, o2 = nd * 10, nd * 10 + nblocks
            with torch.cuda.device(dev):
                _call("az_wgrad_unpack_multi", _p(t), _p(t[o1:o2]), _p(t[o2:]), nd, nblocks, self.stream.cuda_stream)
        self.keep.append(t)

    def join(self):
        if self.joined:
            return
        if self.pending:
            self._flush_pending()
        if self.arena is not None:
            _ARENA_HINT[self.stream.device.index] = self.arena_need
        # (the model's device, not the calling thread's current one: the end-of-backward callback runs outside the
        #  engine's per-node device guard)
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
        self.keep.clear()
        self.arena = None
        self.joined = True
        profiler.joined()

    def ensure_callback(self):
        """called from inside a backward pass, before the first side-stream launch"""
        if not self.callback_set:
            self.callback_set = True
            torch.autograd.Variable._execution_engine.queue_callback(self.join)


class _Gate(torch.autograd.Function):
    """identity on a weight; its backward is held back by the edge to _Tail (module docstring)"""

    @staticmethod
    def forward(ctx, w):
        ctx.set_materialize_grads(False)
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _Tail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, *weights):
        ctx.sink = sink
        ctx.n = len(weights)
        return weights[0].new_zeros(1)

    @staticmethod
    def backward(ctx, _g):
        ctx.sink.join()
        return (None,) * (ctx.n + 1)


def conv_weights(module):
    return [m.weight for m in module.modules()
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose3d))]


def begin(module, like):
    """Start of a forward pass of `module` (all of whose convolutions run on this library's kernels and whose
    first convolution accepts the sink's token): returns a Sink, or None when weight gradients stay in order."""
    if not (torch.is_grad_enabled() and like.is_cuda):
        return None
    weights = conv_weights(module)
    if not weights or not all(w.requires_grad and w.is_leaf for w in weights):
        return None  # (replicas of nn.DataParallel hold non-leaf copies: their gradients flow on through autograd)
    sink = Sink(like.device)
    if sink.stream is not None:
        torch.cuda.current_stream(like.device).wait_stream(sink.stream)  # nothing un-joined survives into this pass
    for w in weights:
        gw = _Gate.apply(w)
        sink.gated[id(w)] = (w, gw)
        sink.owned.add(id(gw))
    sink.token = _Tail.apply(sink, *(g for _, g in sink.gated.values()))
    return sink


class scope:
    """`with scope(sink, *operands):` -- kernels launched inside go to the sink's stream (nothing changes for
    sink None).  Output buffers that the main stream reads after the join are allocated OUTSIDE the scope."""

    def __init__(self, sink, *operands):
        self.sink = sink
        self.operands = operands
        self.ctx = None

    def __enter__(self):
        sink = self.sink
        if sink is not None and sink.live:
            sink.ensure_callback()
            # Operands and outputs live in the MAIN stream's allocator pool and are read / written here on the side
            # stream, so they must outlive the kernel (see _RELEASE above)
            if _RELEASE == "record":
                for t in self.operands:
                    if t is not None:
                        t.record_stream(sink.stream)
            else:
                sink.keep.extend(t for t in self.operands if t is not None)
            sink.stream.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(sink.stream)
            self.ctx.__enter__()
            profiler.side(+1)
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            profiler.side(-1)
            self.ctx.__exit__(*exc)
        return False
