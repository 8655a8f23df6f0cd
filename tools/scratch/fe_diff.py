import copy, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd.nets.psmnet import psmnet_submodule_3 as sub
DEV="cuda:0"
torch.manual_seed(3)
base = sub.FeatureExtraction().to(DEV).to(memory_format=torch.channels_last).train()
left = torch.randn(2, 3, 256, 320, device=DEV).contiguous(memory_format=torch.channels_last)
right = torch.randn(2, 3, 256, 320, device=DEV).contiguous(memory_format=torch.channels_last)
SZ = (int(os.environ.get("FE_H", 256)), int(os.environ.get("FE_W", 320)))
left = torch.randn(2, 3, *SZ, device=DEV).contiguous(memory_format=torch.channels_last)
right = torch.randn(2, 3, *SZ, device=DEV).contiguous(memory_format=torch.channels_last)
GL = torch.randn(2, 32, SZ[0] // 4, SZ[1] // 4, device=DEV); GR = torch.randn(2, 32, SZ[0] // 4, SZ[1] // 4, device=DEV)
def run(mode):
    net = copy.deepcopy(base)
    old = sub.FE2D_BACKEND
    try:
        if mode == "pair":
            sub.FE2D_BACKEND = "fused"; fa, fb = net.forward_pair(left, right)
        elif mode == "seqfused":
            sub.FE2D_BACKEND = "fused"; fa = net(left); fb = net(right)
        else:
            sub.FE2D_BACKEND = "miopen"; fa = net(left); fb = net(right)
    finally:
        sub.FE2D_BACKEND = old
    bufs = {k: v.clone() for k, v in net.named_buffers()}
    ((fa * GL).sum() + (fb * GR).sum()).backward()
    grads = {k: v.grad.clone() for k, v in net.named_parameters()}
    return fa.detach(), fb.detach(), bufs, grads
r = {m: run(m) for m in ("pair", "seqfused", "miopen")}
def d(a, b): return float((a - b).abs().max()), float((a - b).abs().mean()), float(b.abs().max())
for x, y in (("pair", "seqfused"), ("seqfused", "miopen"), ("pair", "miopen")):
    print(x, "vs", y, "fa", d(r[x][0], r[y][0]), "fb", d(r[x][1], r[y][1]))
    worst = max((float((r[x][2][k].float() - r[y][2][k].float()).abs().max()), k) for k in r[x][2])
    print("   worst buffer diff", worst)
    rel = sorted(((float((r[x][3][k] - r[y][3][k]).abs().max() / (r[y][3][k].abs().max() + 1e-12)), k) for k in r[x][3]), reverse=True)[:3]
    print("   worst param-grad rel diffs", rel)
# run miopen twice: its own run-to-run / algorithm noise
a = run("miopen"); print("miopen vs miopen", d(a[0], r["miopen"][0]))
rel = sorted(((float((a[3][k] - r["miopen"][3][k]).abs().max() / (a[3][k].abs().max() + 1e-12)), k) for k in a[3]), reverse=True)[:3]
print("   miopen vs miopen worst param-grad rel diffs", rel)
