import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activezero_amd import ops
B,d,h,w=4,48,136,240
lg=torch.randn(B,1,d,h,w,device="cuda").requires_grad_()
go=torch.randn(B,1,4*h,4*w,device="cuda")
def t(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
def fb():
    out=ops.softargmin(lg); out.backward(go)
print("fwd(no grad) %.3f ms" % t(lambda: ops.softargmin(lg.detach())))
print("fwd+bwd %.3f ms" % t(fb))
