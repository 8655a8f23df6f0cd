#!/bin/bash
# Counter passes of round 2 (each --pmc pass in its own run, with --kernel-trace only):
#   tools/pmc_r02.sh <tag>
# (1) FETCH_SIZE / WRITE_SIZE over tools/kernel_probe.py -> pmc_traffic_b4.json (roofline.traffic)
# (2) SQ issue / wait counters over tools/pmc_aux_probe.py (soft-argmin, patch reprojection)
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o fetch -- python3 $root/tools/kernel_probe.py > $out/pmc_fetch.log 2>&1
echo fetch >> $out/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc -o write -- python3 $root/tools/kernel_probe.py > $out/pmc_write.log 2>&1
echo write >> $out/progress.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o sq1 -- python3 $root/tools/pmc_aux_probe.py > $out/pmc_sq1.log 2>&1
echo sq1 >> $out/progress.txt
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $out/pmc -o sq2 -- python3 $root/tools/pmc_aux_probe.py > $out/pmc_sq2.log 2>&1 || echo "sq2 pass failed (counter set)" >> $out/progress.txt
echo sq2 >> $out/progress.txt
cd $root && python3 tools/pmc_summary.py $out/pmc --out $out/pmc_traffic_b4 > /dev/null
python3 - <<PY
import csv, collections, glob, json, re
t = json.load(open("$out/pmc_traffic_b4.json"))
for k, v in t["kernels"].items():
    print(k, "read %.3f GB write %.3f GB" % (v["read_bytes"] / 1e9, v["write_bytes"] / 1e9))
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for stem in ("sq1", "sq2"):
    for p in glob.glob("$out/pmc/**/%s_counter_collection.csv" % stem, recursive=True):
        for r in csv.DictReader(open(p)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if "softargmin" in name or "patch_reproj" in name:
                rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/sq_aux_summary.md", "w") as f:
    f.write("# SQ counters per launch (mean over 3 launches), B=4, 544x960, D=192, ps=11\n\n")
    for name, cs in rows.items():
        f.write("## %s\n\n| counter | value |\n|---|---|\n" % name)
        for c, v in sorted(cs.items()):
            f.write("| %s | %.4g |\n" % (c, sum(v) / len(v)))
        f.write("\n")
print(open("$out/sq_aux_summary.md").read())
PY
