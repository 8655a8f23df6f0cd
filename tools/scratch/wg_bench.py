import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd import conv3d
dev="cuda:0"
def t(f,n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
x=torch.randn(4,48,136,240,32,device=dev); g=torch.randn(4,48,136,240,32,device=dev)
x64=torch.randn(4,24,68,120,64,device=dev); g64=torch.randn(4,24,68,120,64,device=dev)
print("order", os.environ.get("AZ_WGRAD_ORDER"), "s1 32x32 %.3f ms | s1 64x64(V1) %.3f | s2 64<-32 %.3f" % (
    t(lambda: conv3d._wgrad(g, x, 1, 32, 32, "conv")), t(lambda: conv3d._wgrad(g64, x64, 1, 64, 64, "conv")),
    t(lambda: conv3d._wgrad(g64, x, 2, 64, 32, "conv"))))
