#!/bin/bash
# HBM read traffic of the weight-gradient kernels: FETCH_SIZE pass over tools/bench_s2_family.py --only-s2-wgrad | --only-v0-wgrad
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_s2w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 $root/tools/bench_s2_family.py ${1:---only-s2-wgrad} > $out/fetch.log 2>&1
cd $root
python3 - <<PY
import csv, glob, json, collections
cal = json.load(open("$root/profiles/r04_pmc_traffic_b4.json"))["bytes_per_FETCH_SIZE_unit"]
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/fetch_counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    for d, v in per.items(): acc[name[d]].append(v * cal)
for k, v in sorted(acc.items()):
    if "wgrad" in k: print("%-60s launches %3d  read %.1f MB per launch" % (k[:60], len(v), sum(v) / len(v) / 1e6))
PY
