#!/bin/bash
# SQ / TCP / TCC counter passes over tools/pmc_sq_probe.py (one rocprofv3 run per group, --pmc only
# with --kernel-trace).  usage: tools/pmc_stall_passes.sh <tag>   (AZ_LIB_PATH / AZ_CONV_PRECISION from env)
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/stall_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $out -o a --output-format csv -- python3 $root/tools/${PROBE:-pmc_sq_probe.py} > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES -d $out -o b --output-format csv -- python3 $root/tools/${PROBE:-pmc_sq_probe.py} > $out/b.log 2>&1
python3 - <<PY
import csv, collections, glob
import re
agg = collections.defaultdict(list)
for f in glob.glob("$out/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("gather", "m128", "roll", "wgrad", "conv2d_same", "t2_kernel")):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            agg[(name, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("$tag", k[0], "grid", k[1], k[2], "%.4g" % (sum(agg[k]) / len(agg[k])))
PY
