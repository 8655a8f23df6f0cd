import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd import conv3d
dev = "cuda:0"
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk, ci, co = conv3d._pack_forward(w, 0)
for name, mk in (("random", lambda: torch.randn(4, 48, 136, 240, 32, device=dev)),
                 ("relu(random) (half zeros)", lambda: torch.relu(torch.randn(4, 48, 136, 240, 32, device=dev))),
                 ("zeros", lambda: torch.zeros(4, 48, 136, 240, 32, device=dev))):
    x, g = mk(), mk()
    tc = t(lambda: conv3d._run_gather(x, pk, 0, ci, co, stats=True))
    tw = t(lambda: conv3d._wgrad(g, x, 1, 32, 32, "conv"))
    print(f"{name:28s} conv 32->32 {tc:.3f} ms ({346.5 / tc:.0f} TF)   wgrad 32x32 {tw:.3f} ms ({346.5 / tw:.0f} TF)")
