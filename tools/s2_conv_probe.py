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
for cout in (64, 32):
    w = torch.randn(cout, 32, 3, 3, 3, device=dev) * 0.05
    try:
        pk, ci, co = conv3d._pack_forward(w, conv3d.CONV_S2, A.conv)
        ms = timeit(lambda: conv3d._run_gather(x0, pk, conv3d.CONV_S2, ci, co, A.conv))
        gf = 2.0 * 27 * 32 * cout * 4 * 24 * 68 * 120 / 1e9
        print(f"stride-2 conv 32->{cout}: {ms:.3f} ms  {gf / ms:.1f} TFLOP/s ({gf / ms / 416.7:.2f})")
        ms = timeit(lambda: conv3d._run_gather(x0, pk, conv3d.CONV_S2, ci, co, A.conv, stats=True))
        print(f"   with BN partials: {ms:.3f} ms")
    except Exception as e:
        print("cout", cout, "failed:", e)
# transposed 64 -> 32 (V1 -> V0) and with 32 -> 32
x1 = torch.randn(4, 24, 68, 120, 64, device=dev)
for cin in (64, 32):
    w = torch.randn(cin, 32, 3, 3, 3, device=dev) * 0.05   # ConvTranspose3d weight [cin, cout, ...]
    xin = x1 if cin == 64 else x1[..., :32].contiguous()
    try:
        pk, ci, co = conv3d._pack_forward(w, conv3d.DECONV_S2, A.conv)
        ms = timeit(lambda: conv3d._run_gather(xin, pk, conv3d.DECONV_S2, ci, co, A.conv))
        gf = 2.0 * 27 * cin * 32 * 4 * 24 * 68 * 120 / 1e9
        print(f"transposed {cin}->32: {ms:.3f} ms  {gf / ms:.1f} TFLOP/s ({gf / ms / 416.7:.2f})")
    except Exception as e:
        print("deconv cin", cin, "failed:", e)
