"""The graph construction behind activezero_amd/overlap.py, on CPU tensors: a parameter that is an input of
`_Tail` AND of its own layer receives its gradient (AccumulateGrad, and any hook behind it such as DDP's) only
after `_Tail.backward` has run -- i.e. after the point where the main stream has waited for the side stream --
no matter in which order the layers' own backward nodes hand their weight gradients over."""
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
    def __init__(self, log):
        self.stream, self.token, self.keep, self.armed, self.joined, self.log = None, None, [object()], True, False, log

    def join(self):
        self.log.append(("join",))
        self.keep.clear()
        self.joined = True


def test_every_weight_gradient_is_accumulated_after_the_join():
    log = []
    ws = [torch.nn.Parameter(torch.tensor([float(i + 2)])) for i in range(4)]
    for i, w in enumerate(ws):
        w.register_post_accumulate_grad_hook(lambda p, i=i: log.append(("accumulate", i)))
    sink = _Sink(log)
    sink.token = overlap._Tail.apply(sink, *ws)
    assert sink.token.requires_grad and sink.token.shape == (1,)
    x = torch.ones(1)
    h = _Layer.apply(x, ws[0], sink.token, log, 0)  # the first layer takes the token
    a = _Layer.apply(h, ws[1], None, log, 1)          # two branches that rejoin, as the residual blocks do
    b = _Layer.apply(h, ws[2], None, log, 2)
    out = _Layer.apply(a + b, ws[3], None, log, 3)
    out.sum().backward()
    join_at = log.index(("join",))
    layers = [i for i, e in enumerate(log) if e[0] == "layer"]
    accs = [i for i, e in enumerate(log) if e[0] == "accumulate"]
    assert len(layers) == 4 and len(accs) == 4
    assert max(layers) < join_at < min(accs), log           # all launches, then the join, then every accumulation
    assert log[max(layers)] == ("layer", 0)                  # the token's consumer is the last layer node to run
    assert sink.joined and not sink.keep                     # operands released at the join
    # values: the same graph in plain autograd
    ref = [torch.nn.Parameter(w.detach().clone()) for w in ws]
    h = x * ref[0]
    (((h * ref[1]) + (h * ref[2])) * ref[3]).sum().backward()
    for w, r in zip(ws, ref):
        assert torch.equal(w.grad, r.grad)


def test_unused_token_leaves_the_graph_untouched():
    """If no layer consumes the token, `_Tail` is not reachable from the loss: gradients accumulate as usual
    (and the real sink never arms, so no kernel goes to the side stream)."""
    log = []
    w = torch.nn.Parameter(torch.tensor([3.0]))
    sink = _Sink(log)
    sink.token = overlap._Tail.apply(sink, w)
    _Layer.apply(torch.ones(1), w, None, log, 0).sum().backward()
    assert ("join",) not in log and torch.equal(w.grad, torch.ones(1))


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
