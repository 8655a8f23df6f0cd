#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over tools/kernel_probe.py -> <out>/pmc_traffic_b4.json   usage: tools/pmc_traffic.sh <tag>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o fetch -- python3 $root/tools/kernel_probe.py > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc -o write -- python3 $root/tools/kernel_probe.py > $out/pmc_write.log 2>&1
cd $root && python3 tools/pmc_summary.py $out/pmc --out $out/pmc_traffic_b4
python3 - <<PY
import json
t = json.load(open("$out/pmc_traffic_b4.json"))
for k, v in t["kernels"].items():
    if "conv3d" in k or "bn_apply" in k:
        print(k, "read %.3f GB write %.3f GB" % (v["read_bytes"] / 1e9, v["write_bytes"] / 1e9))
PY
