#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (each --pmc pass in its own run, with --kernel-trace only) over tools/conv2d_roll_probe.py:
# HBM bytes of the batch-walking 2-D kernels against their algorithmic bytes.   usage: tools/pmc_conv2d_roll.sh <tag>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o fetch -- python3 $root/tools/conv2d_roll_probe.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc -o write -- python3 $root/tools/conv2d_roll_probe.py > $out/write.log 2>&1
cd $root
python3 - <<PY
import csv, glob, collections, re
def load(tag, counter):
    per = collections.defaultdict(list)
    for f in glob.glob("$out/pmc/**/%s_counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))].append(float(r["Counter_Value"]))
    return per
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
# calibration as in tools/pmc_summary.py: FETCH_SIZE counts 32-byte... use the documented gfx950 units via a known stream
# kernel is not in this probe, so report raw units x 64 B / x 32 B heuristics are avoided: print raw and the ratio fetch/write
for k in sorted(fe):
    if "conv2d_roll" in k[0] or "conv2d_same" in k[0]:
        f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(len(wr.get(k, [1])), 1)
        print("%-44s grid %-8s launches %3d  FETCH_SIZE %.0f  WRITE_SIZE %.0f" % (k[0][:44], k[1], len(fe[k]), f, w))
PY
