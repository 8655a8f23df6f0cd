import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd import bn2d
DEV="cuda:0"
torch.manual_seed(0)
for C in (32, 64, 128):
    for relu in (False, True):
        for has_res in (False, True):
            for groups in (1, 2):
                n, h, w = 4, 17, 23
                x = (torch.randn(n, C, h, w, device=DEV) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
                res = torch.randn(n, C, h, w, device=DEV).contiguous(memory_format=torch.channels_last) if has_res else None
                gy = torch.randn(n, C, h, w, device=DEV)
                bn_a = torch.nn.BatchNorm2d(C).to(DEV).train(); bn_b = torch.nn.BatchNorm2d(C).to(DEV).train()
                with torch.no_grad():
                    bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.uniform_(-0.5, 0.5)
                    bn_b.load_state_dict(bn_a.state_dict())
                xa = x.clone().requires_grad_(); ra = res.clone().requires_grad_() if has_res else None
                ya = bn2d.bn_act(xa, bn_a, relu, ra, groups)
                ya.backward(gy)
                xb = x.clone().requires_grad_(); rb = res.clone().requires_grad_() if has_res else None
                parts = []
                for g in range(groups):
                    sl = slice(g * n // groups, (g + 1) * n // groups)
                    t = bn_b(xb[sl])
                    if has_res: t = t + rb[sl]
                    parts.append(F.relu(t) if relu else t)
                yb = torch.cat(parts, 0); yb.backward(gy)
                def md(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))
                errs = [md(ya, yb), md(xa.grad, xb.grad), md(bn_a.weight.grad, bn_b.weight.grad), md(bn_a.bias.grad, bn_b.bias.grad),
                        md(bn_a.running_mean, bn_b.running_mean), md(bn_a.running_var, bn_b.running_var)]
                if has_res: errs.append(md(ra.grad, rb.grad))
                ok = max(errs) < 2e-5 and int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked)
                print(C, relu, has_res, groups, "OK" if ok else "FAIL", ["%.1e" % e for e in errs])
