"""TEST INFRASTRUCTURE -- CPU restatement of the RAFT-Stereo ConvGRU update (reference
nets/raft/update.py:19-41), plain torch.  Only tests/ may import this module; the product
(activezero_amd/nets/raft/gru.py) never does.

Pinned: tests/test_oracle_golden.py::test_g12_convgru checks it against G12
(tests/golden/g12_convgru.npz), which tools/make_goldens.py produced by running the reference's own class.
"""
import torch
import torch.nn as nn


class ConvGRUOracle(nn.Module):
    """update.py:20-30: three 3x3 'same' convolutions with bias over [h, x]"""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        pad = kernel_size // 2
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)

    def forward(self, h, cz, cr, cq, *x_list):
        # update.py:32-41
        x = torch.cat(x_list, dim=1)
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz(hx) + cz)        # update gate, context term added before the sigmoid
        r = torch.sigmoid(self.convr(hx) + cr)        # reset gate
        q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q
