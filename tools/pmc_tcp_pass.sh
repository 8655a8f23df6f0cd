#!/bin/bash
# one small TCP/TCC counter group per rocprofv3 run, each bounded by timeout (a 5-counter TCP group aborted)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/tcp_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "group $i: $grp" >> $out/progress.txt
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d $out -o g$i --output-format csv -- python3 $root/tools/pmc_sq_probe.py > $out/g$i.log 2>&1 || { echo "group $i failed rc=$?" >> $out/progress.txt; exit 1; }
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(list)
for f in glob.glob("$out/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("$1", k, "%.4g" % (sum(agg[k]) / len(agg[k])))
PY
