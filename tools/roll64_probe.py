import os, sys, torch
sys.path.insert(0, "/root/repo")
from activezero_amd import conv3d
dev = torch.device("cuda:0"); A = conv3d.DEFAULT_ARITH
def timeit(fn, n=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x0 = torch.randn(4, 48, 136, 240, 32, device=dev)
w0 = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk0, ci0, co0 = conv3d._pack_forward(w0, conv3d.CONV_S1, A.conv)
for _ in range(30): conv3d._run_gather(x0, pk0, conv3d.CONV_S1, ci0, co0, A.conv)
for shape in ((4, 24, 68, 120), (4, 12, 34, 60)):
    x = torch.randn(*shape, 64, device=dev)
    w = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
    gf = 2.0 * 27 * 64 * 64 * x.numel() / 64 / 1e9
    pk = conv3d._pack(w, 64, 64, 64 * 27, 27, False, conv3d.BF16X6)     # the gather kernel's layout
    t_g = timeit(lambda: conv3d._run_gather(x, pk, conv3d.CONV_S1, 64, 64, conv3d.BF16X6))
    t_gs = timeit(lambda: conv3d._run_gather(x, pk, conv3d.CONV_S1, 64, 64, conv3d.BF16X6, stats=True))
    wh = w[:32].contiguous()
    pkr = conv3d._pack(wh, 64, 32, 64 * 27, 27, False, conv3d.BF16X6_R16)
    t_r = timeit(lambda: conv3d._run_gather(x, pkr, conv3d.CONV_S1, 64, 32, conv3d.BF16X6_R16))
    t_rs = timeit(lambda: conv3d._run_gather(x, pkr, conv3d.CONV_S1, 64, 32, conv3d.BF16X6_R16, stats=True))
    print(f"{shape}: gather 64->64 {t_g:.3f} ms ({gf / t_g / 416.7:.2f}), +stats {t_gs:.3f}; roll 64->32 {t_r:.3f} ms x2 = {2 * t_r:.3f} ({gf / (2 * t_r) / 416.7:.2f}), +stats {t_rs:.3f}")
