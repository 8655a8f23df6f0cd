import sys, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from activezero_amd import conv2d
torch.manual_seed(0)
dev = "cuda:0"
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
for pos in (True, False):
    x = torch.rand(2, 128, 32, 64) + (0.0 if pos else -0.5)
    w = (torch.rand(128, 128, 3, 3) + (0.0 if pos else -0.5)) * 0.05
    for exact_bf16 in (True, False):
        xs, ws = (bf(x), bf(w)) if exact_bf16 else (x, w)
        ref = F.conv2d(xs.double(), ws.double(), padding=1)
        y = conv2d.conv_same(xs.to(dev).contiguous(memory_format=torch.channels_last), ws.to(dev), 1).cpu().double()
        t32 = F.conv2d(xs, ws, padding=1).double()
        rel = ((y - ref) / ref.abs().clamp_min(1e-30))
        rel32 = ((t32 - ref) / ref.abs().clamp_min(1e-30))
        sel = ref.abs() > 0.1 * ref.abs().mean()
        print(f"positive={pos} bf16-exact-inputs={exact_bf16}: hip signed mean rel err {rel[sel].mean():+.3e}  mean |rel| {rel[sel].abs().mean():.3e}   torch fp32 signed {rel32[sel].mean():+.3e} |.| {rel32[sel].abs().mean():.3e}")
