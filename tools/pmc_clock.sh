#!/bin/bash
# Effective clock and MFMA-pipe occupancy of the MFMA kernels (MI355X_MICROARCH.md "DVFS give-back"):
#   clock  = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time
#   pipe   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
# one --pmc pass (with --kernel-trace only) per probe.   usage: tools/pmc_clock.sh <tag>
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/clock_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for probe in kernel_probe pmc_sq_probe_conv2d; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $out -o $probe --output-format csv -- python3 $root/tools/$probe.py > $out/$probe.log 2>&1
done
python3 - <<PY
import csv, collections, glob, re
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"] + f.split("/")[-1].split("_kernel")[0]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for f in glob.glob("$out/*_counter_collection.csv"):
    stem = f.split("/")[-1].split("_counter")[0]
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if not any(k in name for k in ("conv3d", "conv2d_same", "conv2d_wgrad", "bn_apply")): continue
        cnt[(name, r["Grid_Size"])][(r["Dispatch_Id"] + stem, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("| kernel | grid | wall us | clock GHz | MFMA pipe busy |")
print("|---|---|---|---|---|")
for (name, grid), d in sorted(cnt.items()):
    ids = sorted({k[0] for k in d})
    ck, pb, wl = [], [], []
    for i in ids[1:] or ids:   # first dispatch of a kernel is cold
        gui = sum(d[(i, "GRBM_GUI_ACTIVE")]); mf = sum(d.get((i, "SQ_VALU_MFMA_BUSY_CYCLES"), [0.0]))
        if i not in dur or gui == 0: continue
        ck.append(gui / 8 / dur[i] / 1e9); pb.append(mf / (1024 * gui / 8)); wl.append(dur[i] * 1e6)
    if ck:
        print("| \`%s\` | %s | %.0f | %.2f | %.2f |" % (name, grid, sum(wl) / len(wl), sum(ck) / len(ck), sum(pb) / len(pb)))
PY
