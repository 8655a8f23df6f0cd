import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd import costconv
from oracle import psmnet_oracle as po
dev = "cuda:0"
torch.manual_seed(0)
for (b, h, w, nd) in ((1, 5, 9, 4), (2, 6, 12, 3), (1, 4, 7, 2), (1, 3, 6, 1), (1, 4, 5, 8)):
    fl = torch.randn(b, 32, h, w, requires_grad=True); fr = torch.randn(b, 32, h, w, requires_grad=True)
    wt = (torch.randn(32, 64, 3, 3, 3) * 0.1).requires_grad_()
    vol = po.build_cost_volume(fl, fr, nd)               # [B,64,nd,h,w]
    ref = F.conv3d(vol, wt, padding=1)                    # [B,32,nd,h,w]
    ct = torch.randn_like(ref)
    gl, gr, gw = torch.autograd.grad(ref, (fl, fr, wt), ct)
    fl2 = fl.detach().to(dev).requires_grad_(); fr2 = fr.detach().to(dev).requires_grad_(); wt2 = wt.detach().to(dev).requires_grad_()
    out = costconv.costvol_conv(fl2, fr2, nd, wt2)       # [B,nd,h,w,32]
    out_n = out.permute(0, 4, 1, 2, 3)
    e = lambda a, r: float((a.cpu() - r).abs().max() / (r.abs().max() + 1e-12))
    g2 = torch.autograd.grad(out_n, (fl2, fr2, wt2), ct.to(dev))
    print((b, h, w, nd), "out %.1e dL %.1e dR %.1e dW %.1e" % (e(out_n, ref), e(g2[0], gl), e(g2[1], gr), e(g2[2], gw)))
