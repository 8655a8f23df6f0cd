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
_PMC_NAMES = {"conv3d_m0_32_32": ("conv3d_roll_kernel<32, 1, 1, false>", "conv3d_roll_kernel<32, 1, 1>", "conv3d_roll_kernel<32, 1, 0>", "conv3d_roll_kernel<32, 1>",
                                  "conv3d_m128_kernel<32, 1, 0>"),
              "dgrad_m0_32_32": ("conv3d_roll_kernel<32, 0, 1, true>", "conv3d_roll_kernel<32, 0, 1, false>", "conv3d_roll_kernel<32, 0, 1>", "conv3d_roll_kernel<32, 0, 0>", "conv3d_roll_kernel<32, 0>",
                                 "conv3d_roll_kernel<32, 2>", "conv3d_m128_kernel<32, 0, 0>"),
              # (a scope of several launches: a tuple of tuples -- the launches' bytes are summed)
              "bn3d_bwd_32": (("bn_bwd_reduce_kernel<32, true>",), ("bn_bwd_apply_kernel<32, true>",)),
              "conv_wgrad_s1_32_32": ("conv3d_wgrad_r16_kernel<1, 1>", "conv3d_wgrad_r16_kernel<1, 0>", "conv3d_wgrad_r16_kernel<1>", "conv3d_wgrad_r16_kernel<0>", "conv3d_wgrad_r16_kernel",
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


# profiler scope (regex) -> the kernel symbol its launches carry in a rocprofv3 kernel trace: a rocprof row sums every
# shape a kernel template runs at, a scope is one shape.  Used for `top_rocprof_kernel*` in the roofline object.
_ROCPROF_FAMILY = [(r"^(conv|deconv)_wgrad_s1_", "conv3d_wgrad_r16_kernel"),
                   (r"^(conv|deconv)_wgrad_s2_", "conv3d_wgrad_s2r16_kernel"),
                   (r"^(conv3d|dgrad)_m0_(32|64)_32$", "conv3d_roll_kernel"),
                   (r"^(conv3d|dgrad)_m2_64_32$", "conv3d_t2roll_kernel"),
                   (r"^(conv3d|dgrad)_m1_", "conv3d_s2roll_kernel / conv3d_gather_kernel (stride 2)"),
                   (r"^(conv3d|dgrad)_m", "conv3d_gather_kernel"),
                   (r"^bn3d_bwd_", "bn_bwd_reduce_kernel + bn_bwd_apply_kernel (3-D)"),
                   (r"^bn2d_bwd", "bn_bwd_reduce_kernel + bn_bwd_apply_kernel (2-D)"),
                   (r"^bn3d_apply_", "bn_apply_kernel"),
                   (r"^conv2d_wgrad", "conv2d_wgrad_*"),
                   (r"^conv2d", "conv2d_*")]


def _family(scope_name):
    import re
    for pat, fam in _ROCPROF_FAMILY:
        if re.search(pat, scope_name):
            return fam
    return scope_name


def roofline(prof, pmc_json=None, clock_json=None, ms_per_step=None, steps=1, solo=None):
    """roofline object for the kernel with the LARGEST TOTAL TIME in the timed region (event-to-event durations on the
    launch stream).  In the backward pass the weight-gradient kernels run on a side stream BESIDE the main stream's
    kernels (overlap.py), so that duration is the one of a kernel sharing the chip: `frac` is the in-step figure,
    `top_kernel_frac_solo` the same launch shape timed alone after the run (`solo`, from bench.py).

    Layout (the driver's parser keeps the leading scalars of an object and drops what follows a nested
    one): the contract's fields first, then every scalar a reader of the line alone needs -- the whole step's matrix rate,
    the top kernel BY ROCPROF NAME (all shapes of one kernel template summed, the way a kernel trace ranks them) with its
    in-step and solo fractions, the largest kernel that ran alone --, and the nested records LAST, under `detail`."""
    if not prof:
        return None
    name, r = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    achieved, peak, peak_basis, unit = _rate(r)
    work_of = lambda v: v["flops"] / 1e12 if v["bound"] == "mfma" else v["bytes"] / 1e9
    out = {"kernel": name, "bound": r["bound"], "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
           "traffic": None}
    # ---- whole step ------------------------------------------------------------------------------------------
    if ms_per_step:
        mf = [v for v in prof.values() if v["bound"] == "mfma"]
        flop_step = sum(v["flops"] * v["launches"] for v in mf) / max(steps, 1)
        votes = {}  # the peak most of the matrix work runs against: flop-weighted vote over the scopes' own peaks
        for v in mf:
            pk = (v.get("peak") or (PEAK["mfma"][0], "fp32 MFMA dense"))
            votes[pk] = votes.get(pk, 0.0) + v["flops"] * v["launches"]
        wpeak, wbasis = max(votes.items(), key=lambda kv: kv[1])[0] if votes else (PEAK["mfma"][0], "fp32 MFMA dense")
        out["whole_step_tflops"] = flop_step / 1e12 / (ms_per_step * 1e-3)
        out["whole_step_frac"] = out["whole_step_tflops"] / wpeak
        out["whole_step_peak"] = wpeak
    # ---- top kernel by rocprof name ----------------------------------------------------------------------------
    fams = {}
    for k, v in prof.items():
        fams.setdefault(_family(k), []).append((k, v))
    fname, members = max(fams.items(), key=lambda kv: sum(v["total_ms"] for _, v in kv[1]))
    f_ms = sum(v["total_ms"] for _, v in members)
    lead_name, lead = max(members, key=lambda kv: kv[1]["total_ms"])
    _, f_peak, f_basis, f_unit = _rate(lead)
    f_work = sum(work_of(v) * v["launches"] for _, v in members)
    out["top_rocprof_kernel"] = fname
    out["top_rocprof_kernel_ms_per_step"] = f_ms / max(steps, 1)
    out["top_rocprof_kernel_frac_in_step"] = f_work / (f_ms * 1e-3) / f_peak
    out["top_rocprof_kernel_frac_solo"] = (work_of(lead) / (solo[lead_name] * 1e-3) / f_peak) if (solo and solo.get(lead_name)) else None
    out["top_rocprof_kernel_bound"] = lead["bound"]
    out["top_rocprof_kernel_lead_scope"] = lead_name
    # ... and the top MATRIX kernel by rocprof name, whatever the overall ranking says (a reader pricing the MFMA side of the step)
    mfams = {f: m for f, m in fams.items() if all(v["bound"] == "mfma" for _, v in m)}
    if mfams:
        mname, mm = max(mfams.items(), key=lambda kv: sum(v["total_ms"] for _, v in kv[1]))
        m_ms = sum(v["total_ms"] for _, v in mm)
        mlead_name, mlead = max(mm, key=lambda kv: kv[1]["total_ms"])
        _, m_peak, _, _ = _rate(mlead)
        out["top_mfma_kernel"] = mname
        out["top_mfma_kernel_ms_per_step"] = m_ms / max(steps, 1)
        out["top_mfma_kernel_frac_in_step"] = sum(work_of(v) * v["launches"] for _, v in mm) / (m_ms * 1e-3) / m_peak
        out["top_mfma_kernel_frac_solo"] = (work_of(mlead) / (solo[mlead_name] * 1e-3) / m_peak) if (solo and solo.get(mlead_name)) else None
        out["top_mfma_kernel_lead_scope"] = mlead_name
    # ---- top scope (one launch shape) ------------------------------------------------------------------------------
    out.update({"top_kernel_by_time": name, "top_kernel_frac_in_step": achieved / peak,
                "top_kernel_ms_in_step": r["avg_ms"], "top_kernel_launches_per_step": r["launches"] / max(steps, 1),
                "top_kernel_frac_solo": None, "top_kernel_ms_solo": None, "traffic_over_algorithmic": None})
    if solo and solo.get(name):
        out["top_kernel_ms_solo"] = solo[name]
        out["top_kernel_frac_solo"] = work_of(r) / (solo[name] * 1e-3) / peak
    # ---- the largest kernel that had the chip to itself ------------------------------------------------------------
    alone = [(k, v) for k, v in prof.items() if not (v.get("side_stream") or v.get("beside_side_stream"))]
    aname, arec = max(alone or list(prof.items()), key=lambda kv: kv[1]["total_ms"])
    a_ach, a_peak, a_basis, _ = _rate(arec)
    out.update({"largest_alone_kernel": aname, "largest_alone_frac": a_ach / a_peak, "largest_alone_ms": arec["avg_ms"],
                "peak_basis": peak_basis, "in_step_overlapped": bool(r.get("side_stream") or r.get("beside_side_stream")),
                "avg_launch_ms": r["avg_ms"], "launches_timed": r["launches"],
                "per_launch_work": r["flops"] if r["bound"] == "mfma" else r["bytes"],
                "selection": "largest total time in the timed region, overlapped kernels included; `frac` is its in-step rate"})
    detail = {"traffic_detail": pmc_traffic(name, r["flops"] if r["bound"] == "mfma" else r["bytes"], pmc_json) if pmc_json else None,
              "held_clock": held_clock(name, clock_json)}
    if detail["traffic_detail"]:
        out["traffic"] = detail["traffic_detail"]["hbm_bytes"]
        alg = r.get("bytes") or 0.0
        if alg > 0:
            out["traffic_over_algorithmic"] = out["traffic"] / alg
    hc = detail["held_clock"]
    if hc:
        out["held_clock_ghz"] = hc.get("clock_ghz")
        out["mfma_pipe_busy"] = hc.get("mfma_pipe_busy")
    if ms_per_step:
        out["whole_step_note"] = (f"algorithmic flops of every MFMA-bound scope of one step (the cost-volume convolution counted in its "
                                  f"factored form) over ms_per_step, against {wbasis}; HBM-bound kernels, Adam and torch glue are in the time")
    if solo:
        detail["solo_ms"] = solo
    detail["families_ms_per_step"] = {f: round(sum(v["total_ms"] for _, v in m) / max(steps, 1), 3) for f, m in
                                      sorted(fams.items(), key=lambda kv: -sum(v["total_ms"] for _, v in kv[1]))[:12]}
    detail["others"] = {k: dict({"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"]},
                                **({"side_stream": True} if v.get("side_stream") else {}),
                                **({"beside_side_stream": True} if v.get("beside_side_stream") else {}))
                        for k, v in prof.items() if k != name}
    out["detail"] = detail
    return out
