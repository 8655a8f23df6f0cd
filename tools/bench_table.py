#!/usr/bin/env python3
"""bench.py's default line (no eager / CPU legs) with its per-scope table: ms per step by kernel family.
    python tools/bench_table.py [n rows] [extra bench.py flags...]"""
import json, subprocess, sys, os
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
extra = [a for a in sys.argv[1:] if not a.isdigit()]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                    "--eager-steps", "0", "--no-stage-bench"] + extra, capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print("%.2f ms/step  %.2f pairs/s  loss %.5f  peak %.1f GB" % (d["ms_per_step"], d["value"], d["loss"], d["peak_mem_gb"]))
rows = [dict(name=k, **v) for k, v in (d["roofline"].get("detail") or d["roofline"])["others"].items()]
top = d["roofline"]
rows.append(dict(name=top["kernel"], avg_ms=top["avg_launch_ms"], launches=top["launches_timed"]))
for x in sorted(rows, key=lambda x: -x.get("avg_ms", 0) * x.get("launches", 0))[:n]:
    print("%-34s %.3f x%-3d = %6.2f %s" % (x["name"], x["avg_ms"], x["launches"] // 5, x["avg_ms"] * x["launches"] / 5,
                                           "side" if x.get("side_stream") else ""))
