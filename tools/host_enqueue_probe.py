#!/usr/bin/env python3
"""Host side of one training step, with N concurrent trainer processes on disjoint cores (the situation on an 8-GPU
node: one process per GPU, all enqueueing ~1150 kernels per step through the same driver).

The step is the bench step (PSMNet fwd + loss + bwd + Adam, every launch the full-size step makes) on a TINY input
(1 x 256 x 320, D = 192: the smallest the SPP branch takes), so that the GPU finishes each kernel long before the host has issued the next one: wall time
per step is then the host's enqueue time -- Python, autograd, ctypes, HIP runtime, kernel driver -- and `thread_time` the
CPU time the enqueueing thread itself burnt.  Compared with the 116 ms of GPU work of the full-size step it gives the
headroom before a rank would become host-bound.  All processes share the ONE GPU of the box (its pool allows 6), which
only adds contention compared with one GPU per process.

    python tools/host_enqueue_probe.py            # N = 1, 2, 4, 6 (child processes are started before any GPU call)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cores, steps, warmup):
    os.sched_setaffinity(0, cores)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from activezero_amd import profiler
    from activezero_amd.nets.psmnet.psmnet_3 import PSMNet
    torch.set_num_threads(1)
    dev = torch.device("cuda:0")
    md = 192
    model = PSMNet(md).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    il, ir, gt = bench.synth_batch(1, 256, 320, md, dev, 7)

    def step():
        opt.zero_grad(set_to_none=True)
        bench.disp_loss(model(il, ir), gt, md).backward()
        opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    profiler.start()
    step()
    torch.cuda.synchronize()
    launches = sum(v.get("launches", 0) for v in profiler.stop().values()) if hasattr(profiler, "stop") else None
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(steps):
        step()
    t_enq, c1 = time.perf_counter(), time.thread_time()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(json.dumps({"enqueue_ms_per_step": 1e3 * (t_enq - t0) / steps, "thread_cpu_ms_per_step": 1e3 * (c1 - c0) / steps,
                      "drain_ms": 1e3 * (t1 - t_enq), "scoped_launches_per_step": launches, "cores": sorted(cores)}))


def main():
    ncores = len(os.sched_getaffinity(0))
    allc = sorted(os.sched_getaffinity(0))
    print(f"host cores available: {ncores}")
    rows = []
    for n in (1, 2, 4, 6):
        per = max(1, ncores // max(n, 8))  # the share a rank has on an 8-GPU node of this pool: cores / 8
        procs = []
        for r in range(n):
            cores = allc[r * per:(r + 1) * per]
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", ",".join(map(str, cores))],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        outs = [json.loads(p.communicate(timeout=600)[0].strip().splitlines()[-1]) for p in procs]
        worst = max(o["enqueue_ms_per_step"] for o in outs)
        cpu = max(o["thread_cpu_ms_per_step"] for o in outs)
        rows.append((n, per, worst, cpu, max(o["drain_ms"] for o in outs)))
        print(f"N={n} processes x {per} core(s): enqueue {worst:.1f} ms/step (slowest process), thread CPU {cpu:.1f} ms/step, "
              f"GPU drain after the last enqueue {rows[-1][4]:.1f} ms")
    print("| processes | cores each | host enqueue ms/step (slowest) | thread CPU ms/step | headroom vs 116 ms of GPU work |")
    print("|---|---|---|---|---|")
    for n, per, w, c, _ in rows:
        print(f"| {n} | {per} | {w:.1f} | {c:.1f} | {116.0 / w:.1f}x |")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child({int(c) for c in sys.argv[2].split(",")}, steps=20, warmup=5)
    else:
        main()
