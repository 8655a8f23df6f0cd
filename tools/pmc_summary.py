#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/kernel_probe.py into per-launch HBM
traffic with the gfx950 corrections of MI355X_MICROARCH.md ("HBM" section):

  * FETCH_SIZE / WRITE_SIZE are reported in KiB-like units -> bytes = value * 1024 ... on
    this ROCm the CSV already carries the raw counter; we therefore CALIBRATE instead of
    trusting a unit: bn_apply streams exactly T bytes in and T bytes out (T = tensor
    bytes), so factor_read = T / FETCH_SIZE(bn_apply), factor_write = T / WRITE_SIZE(bn_apply).
    (The guide's documented case -- FETCH_SIZE reading 1/2 of a 16 B/lane stream -- shows up
    as factor_read ~ 2 x 1024.)
  * every other kernel: bytes = counter * factor.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def load(dirname, stem, counter):
    paths = glob.glob(os.path.join(dirname, "**", f"{stem}_counter_collection.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no {stem}_counter_collection.csv under {dirname}")
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(paths[0])):
        if r["Counter_Name"] == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            per[name].append(float(r["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out", required=True)
    ap.add_argument("--tensor-bytes", type=float, default=4 * 48 * 136 * 240 * 32 * 4.0)
    a = ap.parse_args()
    fetch, write = load(a.dir, "fetch", "FETCH_SIZE"), load(a.dir, "write", "WRITE_SIZE")
    cal = [k for k in fetch if k.startswith("bn_apply_kernel")][0]
    f_read = a.tensor_bytes / (sum(fetch[cal]) / len(fetch[cal]))
    f_write = a.tensor_bytes / (sum(write[cal]) / len(write[cal]))
    out = {"calibration_kernel": cal, "tensor_bytes": a.tensor_bytes, "bytes_per_FETCH_SIZE_unit": f_read,
           "bytes_per_WRITE_SIZE_unit": f_write, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fr = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        wr = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        out["kernels"][k] = {"launches": len(fetch.get(k, [])), "FETCH_SIZE_raw": fr, "WRITE_SIZE_raw": wr,
                             "read_bytes": fr * f_read, "write_bytes": wr * f_write,
                             "hbm_bytes": fr * f_read + wr * f_write}
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
