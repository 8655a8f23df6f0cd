import os, sys, torch
sys.path.insert(0, "/root/repo")
from activezero_amd import _lib, conv2d
from activezero_amd.ops import _call, _p, _stream
dev = torch.device("cuda:0"); lib = _lib.lib()
for (B, H, W, cin, cout) in ((8, 136, 240, 64, 64), (8, 272, 480, 32, 32)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, cin, device=dev).relu()   # post-ReLU activations, as in the extractor
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    pk = torch.empty(int(lib.az_conv2d_roll_packed_floats(cin, cout)), device=dev)
    _call("az_conv2d_roll_pack", _p(pk), _p(w), cin, cout, cin * 9, 9, 0, _stream())
    out = torch.empty(B, H, W, cout, device=dev)
    _call("az_conv2d_roll_fwd", _p(out), _p(x), _p(pk), None, None, None, 0, B, H, W, cin, cout, _stream())
    pk_old = conv2d._pack(w, cin, cout, cin, cout, cin * 9, 9, 3, 3, False)
    os.environ["X"] = "1"
    old = torch.empty(B, H, W, cout, device=dev)
    _call("az_conv2d_fwd", _p(old), _p(x), _p(pk_old), None, None, None, 0, B, H, W, cin, cout, cin, cout, 0, 3, 3, 1, _stream())
    t32 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    sab = float((x.abs().permute(0,3,1,2).double().unsqueeze(0)).mean())  # scale
    denom = float(ref.abs().mean())
    for name, o in (("roll 16x16x32", out), ("K13 32x32x16", old), ("torch fp32", t32)):
        e = (o.double() - ref)
        print(f"{cin}->{cout} {name:14s} signed mean err / mean|ref| {float(e.mean()) / denom:+.3e}   rms err / mean|ref| {float(e.pow(2).mean().sqrt()) / denom:.3e}   per-channel signed mean max {float(e.mean(dim=(0,1,2)).abs().max()) / denom:.3e}")
