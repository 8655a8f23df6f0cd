"""Live per-kernel timing for bench.py's `roofline` object.

`scope(name, flops=..., bytes=...)` brackets ONE kernel launch with HIP events
recorded on torch's current stream -- the stream every az_* kernel is enqueued
on -- so the elapsed time is that kernel's device duration.  Recording only
happens between start() and stop(); outside it a scope costs one `if`.
Algorithmic flops / bytes are the SURVEY.md 8(d) per-unit figures times the
units the launch processes (stated at each call site).
"""
import contextlib

import torch

PEAK = {"mfma": (157.3, "TFLOP/s"),  # dense fp32 MFMA (MI355X_MICROARCH.md, chip table)
        "hbm": (8000.0, "GB/s")}     # HBM3E spec peak

_active = False
_records = {}


def start():
    global _active
    _records.clear()
    _active = True


@contextlib.contextmanager
def scope(name, flops=0.0, bytes=0.0, bound="mfma"):
    if not _active:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _records.setdefault(name, []).append((a, b, float(flops), float(bytes), bound))


def stop():
    global _active
    _active = False
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        ms = [a.elapsed_time(b) for a, b, *_ in recs]
        out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                     "flops": sum(r[2] for r in recs) / len(recs),
                     "bytes": sum(r[3] for r in recs) / len(recs), "bound": recs[0][4]}
    _records.clear()
    return out


def roofline(prof):
    """roofline object for the kernel with the largest total time in the timed region."""
    if not prof:
        return None
    name, r = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    peak, unit = PEAK[r["bound"]]
    work = r["flops"] / 1e12 if r["bound"] == "mfma" else r["bytes"] / 1e9
    achieved = work / (r["avg_ms"] * 1e-3)
    return {"kernel": name, "bound": r["bound"], "achieved": achieved, "peak": peak, "unit": unit,
            "frac": achieved / peak, "traffic": None, "avg_launch_ms": r["avg_ms"],
            "launches_timed": r["launches"],
            "per_launch_work": r["flops"] if r["bound"] == "mfma" else r["bytes"],
            "others": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"]}
                       for k, v in prof.items() if k != name}}
