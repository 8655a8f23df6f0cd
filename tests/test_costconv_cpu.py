"""CPU (no GPU): the algebra behind activezero_amd/costconv.py.  The merged 2-D kernels that the host
code builds from the Conv3d weight, convolved with the feature maps on the CPU and assembled by a
plain-torch restatement of csrc/az_costconv.hip's index rule, must equal conv3d(concat cost volume)
from the oracle -- including the staircase mask, both depth borders, the right image edge and the
degenerate depths D = 1, 2."""
import pytest
import torch
import torch.nn.functional as F

from activezero_amd import costconv
from oracle import psmnet_oracle as po
from tests._weights import seeded


def _cls(d, nd):  # csrc/az_costconv.hip cc_class
    return 0 if (nd == 1 or d == 0) else ((1 if nd == 2 else 2) if d == nd - 1 else 1)


def _assemble(fb, fe, g, nd):
    """out[b,:,d,y,x] = F_{c(d),min(x-d,2)}[y,x] + G_{c(d),[x=W-1]}[y,x-d] for x-d >= -2, else 0 (NCHW maps)."""
    b, _, h, w = fb.shape
    out = torch.zeros(b, 32, nd, h, w)
    for d in range(nd):
        c = _cls(d, nd)
        for x in range(w):
            delta = x - d
            if delta < -2:
                continue
            if delta >= 2:
                f = fb[:, c * 32:(c + 1) * 32, :, x]
            else:
                k = c * 4 + (delta + 2)
                f = fe[:, k * 32:(k + 1) * 32, :, x]
            k = c * 2 + (1 if x == w - 1 else 0)
            out[:, :, d, :, x] = f + g[:, k * 32:(k + 1) * 32, :, delta + 2]
    return out


@pytest.mark.parametrize("dims", [(1, 4, 9, 4), (2, 3, 7, 3), (1, 3, 6, 2), (1, 2, 5, 1), (1, 3, 4, 7)])
def test_factored_cost_volume_convolution(dims):
    b, h, w, nd = dims
    fl, fr = seeded((b, 32, h, w), 1), seeded((b, 32, h, w), 2)
    wt = seeded((32, 64, 3, 3, 3), 3, -0.1, 0.1)
    ref = F.conv3d(po.build_cost_volume(fl, fr, nd), wt, padding=1)
    kb, ke, kr = costconv._merged_kernels(wt, nd)
    assert kb.shape[0] == costconv.num_classes(nd) * 32 and ke.shape[0] == 4 * kb.shape[0]
    xe = min(w, nd + 1)
    fb = F.conv2d(fl, kb, padding=1)
    fe = F.conv2d(fl[..., :min(w, xe + 1)], ke, padding=1)[..., :xe]
    g = F.conv2d(F.pad(fr, (2, 0)), kr, padding=(1, 2))
    out = _assemble(fb, fe, g, nd)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5), float((out - ref).abs().max())
