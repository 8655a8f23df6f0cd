#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over bench.py's TIMED region only.

rocprofv3 traces the whole process (MIOpen's find phase, warm-up steps, ...).
The cost-volume assemble kernel `costconv_assemble_fwd_kernel`
(dres0[0], train mode) runs exactly once per step, so its (warmup+1)-th dispatch
marks the start of the timed steps.

    python tools/prof_summary.py <kernel_trace.csv> --warmup W --out profiles/<name>
writes <name>_kernel_stats.csv and <name>_summary.md
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--marker", default="costconv_assemble_fwd_kernel")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
    if len(marks) <= a.warmup:
        raise SystemExit(f"marker kernel seen {len(marks)} times, need > {a.warmup}")
    t0 = marks[a.warmup]
    steps = len(marks) - a.warmup
    # a step starts with the 2-D feature extractor, before the marker: back up to the
    # previous optimizer tail by taking the gap just before the marker's step as boundary
    # -> simply start at the first dispatch after the previous step's last kernel.
    prev_end = max((int(r["End_Timestamp"]) for r in rows if int(r["End_Timestamp"]) < marks[a.warmup - 1] if a.warmup > 0), default=0)
    sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
    # include the part of the step that precedes the marker (feature extraction): find
    # the end of the previous step = last Adam kernel before t0
    adam = [int(r["End_Timestamp"]) for r in rows
            if int(r["End_Timestamp"]) < t0 and "multi_tensor_apply" in r["Kernel_Name"]]
    if adam:
        t0 = adam[-1]
        sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
    agg = collections.OrderedDict()
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        c = agg.setdefault(k, [0, 0, 10**18, 0])
        c[0] += 1; c[1] += d; c[2] = min(c[2], d); c[3] = max(c[3], d)
    total = sum(c[1] for c in agg.values())
    wall = max(int(r["End_Timestamp"]) for r in sel) - t0
    # kernels of the two streams run side by side (overlap.py): time with at least one / at least two in flight
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in sel] + [(int(r["End_Timestamp"]), -1) for r in sel])
    busy = both = depth = 0
    last = ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(a.out + "_kernel_stats.csv", "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for k, (n, tot, mn, mx) in items:
            f.write(f"\"{k}\",{n},{tot},{tot / n:.1f},{100.0 * tot / total:.2f},{mn},{mx}\n")
    with open(a.out + "_summary.md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace summary, timed region only ({steps} steps)\n\n")
        if a.note:
            f.write(a.note + "\n\n")
        f.write(f"wall {wall / 1e6:.1f} ms ({wall / 1e6 / steps:.1f} ms per step); at least one kernel in flight "
                f"{busy / 1e6 / steps:.1f} ms per step, two or more {both / 1e6 / steps:.1f} ms per step; sum of kernel "
                f"durations {total / 1e6 / steps:.1f} ms per step (kernels that share the chip run longer each)\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for k, (n, tot, mn, mx) in items[:45]:
            f.write(f"| `{k}` | {n} | {tot / 1e6:.2f} | {tot / n / 1e3:.1f} | {100.0 * tot / total:.1f} |\n")
    print(open(a.out + "_summary.md").read())


if __name__ == "__main__":
    main()
