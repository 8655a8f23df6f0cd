#!/bin/bash
# Counter passes of round 5 over tools/kernel_probe.py (each --pmc pass in its own run, --kernel-trace only):
#   FETCH_SIZE / WRITE_SIZE -> r05_pmc_traffic_b4.json (roofline.traffic), clock / MFMA pipe -> r05_pmc_clock_b4.json
#   usage: tools/pmc_r05.sh <tag>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o fetch -- python3 $root/tools/kernel_probe.py > $out/pmc_fetch.log 2>&1
echo fetch >> $out/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc -o write -- python3 $root/tools/kernel_probe.py > $out/pmc_write.log 2>&1
echo write >> $out/progress.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $out/pmc -o clock --output-format csv -- python3 $root/tools/kernel_probe.py > $out/pmc_clock.log 2>&1
echo clock >> $out/progress.txt
cd $root && python3 tools/pmc_summary.py $out/pmc --out $out/r05_pmc_traffic_b4 > /dev/null
python3 - <<PY
import csv, collections, glob, json, re
dur = {}
for f in glob.glob("$out/pmc/**/clock_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for f in glob.glob("$out/pmc/**/clock_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
per = collections.defaultdict(list)
for d, c in cnt.items():
    if d in dur and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        per[names[d]].append((c["GRBM_GUI_ACTIVE"] / 8 / dur[d] / 1e9, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * c["GRBM_GUI_ACTIVE"] / 8), dur[d] * 1e6))
outj = {"source": "tools/pmc_r05.sh over tools/kernel_probe.py (B=4, V0 = 48x136x240, 32 channels): clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall; pipe = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); first dispatch of a kernel dropped (cold)",
        "nominal_clock_ghz": 2.4, "kernels": {}}
for k, v in per.items():
    v = v[1:] or v
    outj["kernels"][k] = {"clock_ghz": round(sum(x[0] for x in v) / len(v), 3), "mfma_pipe_busy": round(sum(x[1] for x in v) / len(v), 3),
                          "wall_us": round(sum(x[2] for x in v) / len(v), 1)}
json.dump(outj, open("$out/r05_pmc_clock_b4.json", "w"), indent=1)
t = json.load(open("$out/r05_pmc_traffic_b4.json"))
for k, v in t["kernels"].items():
    print(k, "read %.3f GB write %.3f GB" % (v["read_bytes"] / 1e9, v["write_bytes"] / 1e9), outj["kernels"].get(k))
PY
