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
_side_depth = 0  # measurement tooling only (bench.py, one thread)
_side_pending = False  # side-stream work launched and not yet joined: main-stream kernels share the chip


def side(delta):
    global _side_depth, _side_pending
    _side_depth += delta
    _side_pending = True


def joined():
    global _side_pending
    _side_pending = False


def start():
    global _active, _side_depth, _side_pending
    _records.clear()
    _side_depth, _side_pending = 0, False  # (a backward pass that died before its join leaves them set)
    _active = True


@contextlib.contextmanager
def scope(name, flops=0.0, bytes=0.0, bound="mfma", peak=None):
    if not _active:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    side = _side_depth > 0  # inside overlap.scope: a weight-gradient kernel on the side stream
    beside = _side_pending and not side  # a main-stream kernel while side-stream kernels are in flight
    a.record()
    try:
        yield
    finally:
        b.record()
        _records.setdefault(name, []).append((a, b, float(flops), float(bytes), bound, peak, side, beside))


def stop():
    global _active
    _active = False
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        ms = [a.elapsed_time(b) for a, b, *_ in recs]
        out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                     "flops": sum(r[2] for r in recs) / len(recs),
                     "bytes": sum(r[3] for r in recs) / len(recs), "bound": recs[0][4],
                     "peak": recs[0][5], "side_stream": any(r[6] for r in recs),
                     "beside_side_stream": any(r[7] for r in recs)}
    _records.clear()
    return out


# profiler scope name -> kernel symbol in the rocprofv3 PMC summary (newest kernel first)
_PMC_NAMES = {"conv3d_m0_32_32": ("conv3d_roll_kernel<32, 1, 1>", "conv3d_roll_kernel<32, 1, 0>", "conv3d_roll_kernel<32, 1>",
                                  "conv3d_m128_kernel<32, 1, 0>"),
              "dgrad_m0_32_32": ("conv3d_roll_kernel<32, 0, 1>", "conv3d_roll_kernel<32, 0, 0>", "conv3d_roll_kernel<32, 0>",
                                 "conv3d_roll_kernel<32, 2>", "conv3d_m128_kernel<32, 0, 0>"),
              # (a scope of several launches: a tuple of tuples -- the launches' bytes are summed)
              "bn3d_bwd_32": (("bn_bwd_reduce_kernel<32, true>",), ("bn_bwd_apply_kernel<32, true>",)),
              "conv_wgrad_s1_32_32": ("conv3d_wgrad_r16_kernel<1>", "conv3d_wgrad_r16_kernel<0>", "conv3d_wgrad_r16_kernel",
                                      "conv3d_wgrad_x6_kernel<32, 32, 1>", "conv3d_wgrad_kernel<32, 32, 1>")}


def pmc_traffic(scope_name, per_launch_work, pmc_json):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (tools/kernel_probe.py + tools/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes,
    unit-calibrated on a pure streaming kernel).  Only returned when the probe ran the same
    launch shape (same algorithmic flops per launch); otherwise None."""
    import json
    import os

    if not os.path.exists(pmc_json) or scope_name not in _PMC_NAMES:
        return None
    data = json.load(open(pmc_json))
    syms = _PMC_NAMES[scope_name]
    if syms and isinstance(syms[0], tuple):  # several launches per scope (BatchNorm backward: reduce + apply)
        parts = []
        for alts in syms:
            hit = next((data["kernels"][a] for a in alts if a in data["kernels"]), None)
            if hit is None:
                return None
            parts.append(hit)
        k = {f: sum(p[f] for p in parts) for f in ("hbm_bytes", "read_bytes", "write_bytes")}
        probe_work = 5.0 * data["tensor_bytes"]  # five tensor passes: (dy, raw) twice in, dx out
    else:
        k = None
        for sym in syms:
            k = k or data["kernels"].get(sym)
        probe_work = 2.0 * 27 * 32 * 32 * data["tensor_bytes"] / (4 * 32)
    if not k or abs(probe_work - per_launch_work) > 1e-6 * probe_work:
        return None
    return {"hbm_bytes": k["hbm_bytes"], "read_bytes": k["read_bytes"], "write_bytes": k["write_bytes"],
            "carried_from": "profiles/" + os.path.basename(pmc_json),
            "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kernel_probe.py at this launch "
                    "shape on another box, units calibrated on bn_apply (a pure stream); not measured in this run"}


def held_clock(scope_name, clock_json):
    """Clock and MFMA-pipe occupancy of the kernel from the committed counter pass (tools/pmc_clock.sh:
    GRBM_GUI_ACTIVE / 8 / wall, SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)); None when absent."""
    import json
    import os

    if not clock_json or not os.path.exists(clock_json) or scope_name not in _PMC_NAMES:
        return None
    data = json.load(open(clock_json))["kernels"]
    for sym in _PMC_NAMES[scope_name]:
        if sym in data:
            return dict(data[sym], kernel=sym, carried_from="profiles/" + os.path.basename(clock_json),
                        note="counter pass (tools/pmc_clock.sh) on another box; not measured in this run")
    return None


def _rate(r):
    peak, unit = PEAK[r["bound"]]
    basis = "fp32 MFMA dense" if r["bound"] == "mfma" else "HBM3E spec"
    if r.get("peak"):
        peak, basis = r["peak"]
    work = r["flops"] / 1e12 if r["bound"] == "mfma" else r["bytes"] / 1e9
    return work / (r["avg_ms"] * 1e-3), peak, basis, unit


def roofline(prof, pmc_json=None, clock_json=None, ms_per_step=None, steps=1, solo=None):
    """roofline object for the kernel with the LARGEST TOTAL TIME in the timed region (event-to-event durations on the
    launch stream).  In the backward pass the weight-gradient kernels run on a side stream BESIDE the main stream's
    kernels (overlap.py), so that duration is the one of a kernel sharing the chip: `frac` is the in-step figure,
    `top_kernel_frac_solo` the same launch shape timed alone after the run (`solo`, from bench.py).  Scalars at the top
    level (they survive parsers that drop nested objects): top_kernel_by_time, top_kernel_frac_in_step,
    top_kernel_frac_solo, whole_step_tflops / whole_step_frac (all MFMA-bound scopes' algorithmic flops over the step
    time, against the peak of the arithmetic most of them run in)."""
    if not prof:
        return None
    name, r = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    achieved, peak, peak_basis, unit = _rate(r)
    alone = [(k, v) for k, v in prof.items() if not (v.get("side_stream") or v.get("beside_side_stream"))]
    aname, arec = max(alone or list(prof.items()), key=lambda kv: kv[1]["total_ms"])
    a_ach, a_peak, a_basis, _ = _rate(arec)
    out = {"kernel": name, "bound": r["bound"], "achieved": achieved, "peak": peak, "peak_basis": peak_basis, "unit": unit,
           "frac": achieved / peak,
           "selection": "largest total time in the timed region, overlapped kernels included; `frac` is its in-step rate",
           "in_step_overlapped": bool(r.get("side_stream") or r.get("beside_side_stream")),
           "top_kernel_by_time": name, "top_kernel_frac_in_step": achieved / peak,
           "top_kernel_ms_in_step": r["avg_ms"], "top_kernel_launches_per_step": r["launches"] / max(steps, 1),
           "top_kernel_frac_solo": None, "top_kernel_ms_solo": None,
           "traffic": None, "traffic_over_algorithmic": None,  # scalars: HBM bytes per launch (PMC passes), and over SURVEY 8d's bytes
           "traffic_detail": pmc_traffic(name, r["flops"] if r["bound"] == "mfma" else r["bytes"], pmc_json) if pmc_json else None,
           "held_clock": held_clock(name, clock_json),
           "avg_launch_ms": r["avg_ms"], "launches_timed": r["launches"],
           "per_launch_work": r["flops"] if r["bound"] == "mfma" else r["bytes"],
           "largest_alone": {"kernel": aname, "achieved": a_ach, "peak": a_peak, "peak_basis": a_basis, "frac": a_ach / a_peak,
                             "avg_ms": arec["avg_ms"], "launches": arec["launches"],
                             "note": "largest total time among the kernels that ran with the chip to themselves"}}
    if out["traffic_detail"]:
        out["traffic"] = out["traffic_detail"]["hbm_bytes"]
        alg = r.get("bytes") or 0.0
        if alg > 0:
            out["traffic_over_algorithmic"] = out["traffic"] / alg
    hc = out.get("held_clock")
    if hc:  # (scalars for readers that drop nested objects)
        out["held_clock_ghz"] = hc.get("clock_ghz")
        out["mfma_pipe_busy"] = hc.get("mfma_pipe_busy")
    if solo and solo.get(name):
        out["top_kernel_ms_solo"] = solo[name]
        work = r["flops"] / 1e12 if r["bound"] == "mfma" else r["bytes"] / 1e9
        out["top_kernel_frac_solo"] = work / (solo[name] * 1e-3) / peak
    if solo:
        out["solo_ms"] = solo
    if ms_per_step:
        mf = [v for v in prof.values() if v["bound"] == "mfma"]
        flop_step = sum(v["flops"] * v["launches"] for v in mf) / max(steps, 1)
        # the peak most of the matrix work runs against: flop-weighted vote over the scopes' own peaks
        votes = {}
        for v in mf:
            pk = (v.get("peak") or (PEAK["mfma"][0], "fp32 MFMA dense"))
            votes[pk] = votes.get(pk, 0.0) + v["flops"] * v["launches"]
        wpeak, wbasis = max(votes.items(), key=lambda kv: kv[1])[0] if votes else (PEAK["mfma"][0], "fp32 MFMA dense")
        out["whole_step_tflops"] = flop_step / 1e12 / (ms_per_step * 1e-3)
        out["whole_step_frac"] = out["whole_step_tflops"] / wpeak
        out["whole_step_peak"] = wpeak
        out["whole_step_peak_basis"] = wbasis
        out["whole_step_mfma_flop"] = flop_step
        out["whole_step_note"] = ("algorithmic flops of every MFMA-bound scope of one step (the cost-volume convolution counted "
                                  "in its factored form) over ms_per_step; HBM-bound kernels, Adam and torch glue are in the time")
    out["others"] = {k: dict({"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"]},
                             **({"side_stream": True} if v.get("side_stream") else {}),
                             **({"beside_side_stream": True} if v.get("beside_side_stream") else {}))
                     for k, v in prof.items() if k != name}
    return out
