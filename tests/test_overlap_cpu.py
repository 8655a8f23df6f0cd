"""The graph construction behind activezero_amd/overlap.py, on CPU tensors: a parameter whose layer uses its
gated alias (`_Gate`, itself an input of `_Tail`) receives its gradient (AccumulateGrad, and any hook behind it
such as DDP's) only after `_Tail.backward` has run -- i.e. after the point where the main stream has waited for
the side stream -- no matter in which order the layers' own backward nodes hand their weight gradients over, and
also when two forward passes are back-propagated together."""
import torch

from activezero_amd import overlap


class _Layer(torch.autograd.Function):
    """y = x * w (+ token); its backward logs when it runs and returns the weight gradient at once, as the
    convolution nodes do (the real ones have merely LAUNCHED the kernel on the side stream by then)."""

    @staticmethod
    def forward(ctx, x, w, token, log, name):
        ctx.save_for_backward(x, w)
        ctx.log, ctx.name, ctx.has_token = log, name, token is not None
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        ctx.log.append(("layer", ctx.name))
        return g * w, (g * x).sum().reshape(w.shape), (g.new_zeros(1) if ctx.has_token else None), None, None


class _Sink(overlap.Sink):
    def __init__(self, log, tag=0):
        self.stream, self.token, self.keep, self.armed, self.joined = None, None, [object()], True, False
        self.gated, self.owned, self.log, self.tag = {}, set(), log, tag
        self.uses, self.disarmed, self.callback_set = {}, False, False

    def join(self):
        self.log.append(("join", self.tag))
        self.keep.clear()
        self.joined = True


def _begin(ws, log, tag=0):
    """overlap.begin() with the stub sink (no GPU here)"""
    sink = _Sink(log, tag)
    for w in ws:
        g = overlap._Gate.apply(w)
        sink.gated[id(w)] = (w, g)
        sink.owned.add(id(g))
    sink.token = overlap._Tail.apply(sink, *(g for _, g in sink.gated.values()))
    return sink


def _net(x, ws, sink, log, tag=0):
    g = sink.weight
    assert all(sink.owns(g(w)) and not sink.owns(w) for w in ws)
    h = _Layer.apply(x, g(ws[0]), sink.token, log, (tag, 0))  # the first layer takes the token
    a = _Layer.apply(h, g(ws[1]), None, log, (tag, 1))          # two branches that rejoin, as the residual blocks do
    b = _Layer.apply(h, g(ws[2]), None, log, (tag, 2))
    return _Layer.apply(a + b, g(ws[3]), None, log, (tag, 3))


def _plain(x, ws):
    h = x * ws[0]
    return ((h * ws[1]) + (h * ws[2])) * ws[3]


def test_every_weight_gradient_is_accumulated_after_the_join():
    log = []
    ws = [torch.nn.Parameter(torch.tensor([float(i + 2)])) for i in range(4)]
    for i, w in enumerate(ws):
        w.register_post_accumulate_grad_hook(lambda p, i=i: log.append(("accumulate", i)))
    sink = _begin(ws, log)
    assert sink.token.requires_grad and sink.token.shape == (1,)
    x = torch.ones(1)
    _net(x, ws, sink, log).sum().backward()
    join_at = log.index(("join", 0))
    layers = [i for i, e in enumerate(log) if e[0] == "layer"]
    accs = [i for i, e in enumerate(log) if e[0] == "accumulate"]
    assert len(layers) == 4 and len(accs) == 4
    assert max(layers) < join_at < min(accs), log           # all launches, then the join, then every accumulation
    assert log[max(layers)] == ("layer", (0, 0))             # the token's consumer is the last layer node to run
    assert sink.joined and not sink.keep                     # operands released at the join
    ref = [torch.nn.Parameter(w.detach().clone()) for w in ws]
    _plain(x, ref).sum().backward()
    for w, r in zip(ws, ref):
        assert torch.equal(w.grad, r.grad)


def test_two_passes_backpropagated_together_each_wait_for_their_own_join():
    """Gradient accumulation over two forward passes with ONE backward: a weight then has two real gradients.
    Each passes its own gate after its own join, so the engine's early `grad1 + grad2` in the weight's input
    buffer only ever sees joined tensors."""
    log = []
    ws = [torch.nn.Parameter(torch.tensor([float(i + 2)])) for i in range(4)]
    gate_log = []
    real_gate_bwd = overlap._Gate.backward
    x1, x2 = torch.ones(1), torch.full((1,), 3.0)
    s1 = _begin(ws, log, 1)
    y1 = _net(x1, ws, s1, log, 1)
    s2 = _begin(ws, log, 2)
    y2 = _net(x2, ws, s2, log, 2)
    try:
        overlap._Gate.backward = staticmethod(lambda ctx, g: (gate_log.append(tuple(e for e in log if e[0] == "join")), g)[1])
        (y1.sum() + y2.sum()).backward()
    finally:
        overlap._Gate.backward = real_gate_bwd
    assert ("join", 1) in log and ("join", 2) in log and s1.joined and s2.joined
    # every gate ran after at least its own pass's join; 8 gates (4 weights x 2 passes) ran in all
    assert len(gate_log) == 8 and all(len(j) >= 1 for j in gate_log)
    for tag in (1, 2):
        last_layer = max(i for i, e in enumerate(log) if e[0] == "layer" and e[1][0] == tag)
        assert last_layer < log.index(("join", tag))
    ref = [torch.nn.Parameter(w.detach().clone()) for w in ws]
    (_plain(x1, ref).sum() + _plain(x2, ref).sum()).backward()
    for w, r in zip(ws, ref):
        assert torch.equal(w.grad, r.grad)


def test_unused_token_leaves_the_graph_untouched():
    """If no layer consumes the token, `_Tail` is not reachable from the loss: gradients accumulate as usual
    (and the real sink never arms, so no kernel goes to the side stream)."""
    log = []
    w = torch.nn.Parameter(torch.tensor([3.0]))
    sink = _begin([w], log)
    _Layer.apply(torch.ones(1), sink.weight(w), None, log, 0).sum().backward()
    assert not any(e[0] == "join" for e in log) and torch.equal(w.grad, torch.ones(1))


def test_begin_declines_without_grad_mode_cpu_tensors_or_with_frozen_weights():
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv3d(4, 4, 3))
    assert overlap.begin(m, torch.zeros(1)) is None            # CPU tensor
    assert len(overlap.conv_weights(m)) == 2
    real = overlap.Sink.__init__
    try:
        overlap.Sink.__init__ = lambda self, device: _Sink.__init__(self, [])  # no GPU here: stub the stream

        class _Cuda:
            is_cuda, device = True, "cuda:0"
        with torch.no_grad():
            assert overlap.begin(m, _Cuda()) is None           # no grad mode
        assert overlap.begin(m, _Cuda()) is not None
        m[0].weight.requires_grad_(False)
        assert overlap.begin(m, _Cuda()) is None               # a frozen convolution weight
    finally:
        overlap.Sink.__init__ = real


def test_weight_requested_twice_disarms_the_sink():
    """A gate with two REAL gradient edges would be summed by the engine before the join: the sink then keeps
    every weight gradient of the pass in order (overlap.py, backstops)."""
    log = []
    ws = [torch.nn.Parameter(torch.tensor([float(i + 2)])) for i in range(2)]
    sink = _begin(ws, log)
    assert sink.live
    a = sink.weight(ws[0])
    assert sink.live and sink.weight(ws[1]) is not ws[1]
    assert sink.weight(ws[0]) is a        # the same alias again ...
    assert sink.disarmed and not sink.live  # ... and nothing of this pass goes to the side stream any more
    assert sink.weight(torch.ones(1)) is not None  # tensors that are not gated pass through untouched


def test_join_is_idempotent_and_registered_once():
    calls = []
    sink = overlap.Sink.__new__(overlap.Sink)
    sink.stream, sink.keep, sink.joined, sink.callback_set = None, [1], False, False

    class _Eng:
        @staticmethod
        def queue_callback(cb):
            calls.append(cb)
    real = torch.autograd.Variable._execution_engine
    try:
        torch.autograd.Variable._execution_engine = _Eng
        sink.ensure_callback()
        sink.ensure_callback()
    finally:
        torch.autograd.Variable._execution_engine = real
    assert len(calls) == 1
    sink.joined = True
    sink.join()  # already joined: returns without touching the (absent) stream
    assert sink.keep == [1]
