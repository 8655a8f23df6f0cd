#!/usr/bin/env python3
"""What taking max |y| inside the BatchNorm apply kernel costs, alone: az_bn3d_apply with and without y_amax."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for shape in ((4, 24, 68, 120, 64), (4, 48, 136, 240, 32), (8, 136, 240, 64)):
    C = shape[-1]
    x = torch.randn(*shape, device=dev); y = torch.empty_like(x)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    am = torch.zeros(1024, device=dev)
    nv = x.numel() // C
    t0 = timeit(lambda: ops._call("az_bn3d_apply", y.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, nv, C, None, ops._stream()))
    t1 = timeit(lambda: ops._call("az_bn3d_apply", y.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, nv, C, am.data_ptr(), ops._stream()))
    def fresh():
        am.zero_()
        ops._call("az_bn3d_apply", y.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, nv, C, am.data_ptr(), ops._stream())
    t2 = timeit(fresh)
    pool = torch.zeros(64, 1024, device=dev)  # a fresh all-zero array per launch, zeroed outside the timed region
    it = [0]
    def pooled():
        ops._call("az_bn3d_apply", y.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, nv, C, pool[it[0] % 64].data_ptr(), ops._stream())
        it[0] += 1
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(25): pooled()
    b.record(); torch.cuda.synchronize()
    t3 = a.elapsed_time(b) / 25
    print(f"{shape}: no amax {t0*1e3:.1f} us, amax (slots already hold the max) {t1*1e3:.1f} us, amax from zero + a fill kernel {t2*1e3:.1f} us, amax from a pre-zeroed array {t3*1e3:.1f} us")
