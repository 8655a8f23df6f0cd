"""Time the RAFT-Stereo ConvGRU update (activezero_amd.nets.raft.gru) against the same formula in eager PyTorch
under autocast(bfloat16) on the finest GRU level of the reference's default size: hidden 128, input 256,
[B,*,136,240] (1/4 of 544x960; raft_stereo.py:138-172 runs it 22 times per step)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activezero_amd.nets.raft.gru import ConvGRU  # noqa: E402


def eager(mod, h, cz, cr, cq, x):
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(mod.convz(hx) + cz)
    r = torch.sigmoid(mod.convr(hx) + cr)
    q = torch.tanh(mod.convq(torch.cat([r * h, x], 1)) + cq)
    return (1 - z) * h + z * q


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    hid, cin, hh, ww = 128, 256, 136, 240
    mod = ConvGRU(hid, cin).to(dev)
    h = torch.tanh(torch.randn(args.batch, hid, hh, ww, device=dev))
    cz, cr, cq = (torch.randn(args.batch, hid, hh, ww, device=dev) for _ in range(3))
    x = torch.randn(args.batch, cin, hh, ww, device=dev)
    flops = 3 * 18.0 * (hid + cin) * hid * args.batch * hh * ww
    with torch.no_grad():
        t_hip = timed(lambda: mod(h, cz, cr, cq, x), args.iters)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t_amp = timed(lambda: eager(mod, h, cz, cr, cq, x), args.iters)
        t_f32 = timed(lambda: eager(mod, h, cz, cr, cq, x), args.iters)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            err_amp = (eager(mod, h, cz, cr, cq, x).float() - eager(mod.double(), h.double(), cz.double(), cr.double(),
                                                                    cq.double(), x.double())).abs().max().item()
        mod.float()
        err_hip = (mod(h, cz, cr, cq, x).double() - eager(mod.double(), h.double(), cz.double(), cr.double(),
                                                          cq.double(), x.double())).abs().max().item()
    print(f"ConvGRU update B={args.batch} [{hid}+{cin}]x{hh}x{ww}: {flops / 1e9:.1f} GFLOP")
    print(f"  HIP bf16 (2 launches + layout glue) {t_hip:8.3f} ms  {flops / t_hip / 1e9:7.1f} TFLOP/s   max err vs fp64 {err_hip:.2e}")
    print(f"  eager autocast(bf16)                {t_amp:8.3f} ms  {flops / t_amp / 1e9:7.1f} TFLOP/s   max err vs fp64 {err_amp:.2e}")
    print(f"  eager fp32                          {t_f32:8.3f} ms  {flops / t_f32 / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
