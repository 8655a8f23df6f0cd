#!/bin/bash
# the distinct kernel names of one short bench run (for tools/isa_lint_torch.py --trace <names.txt>):
#   tools/trace_kernels.sh <tag> <bench.py flags...>     -> gpurun_out/<tag>_kernel_names.txt
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --eager-steps 0 --no-stage-bench "$@" > $out/trace_bench.log 2>&1
python3 - <<PY
import csv, glob
names = set()
for f in glob.glob("$out/trace/**/t_kernel_trace.csv", recursive=True):
    names |= {r["Kernel_Name"] for r in csv.DictReader(open(f))}
open("$root/gpurun_out/${tag}_kernel_names.txt", "w").write("\n".join(sorted(names)) + "\n")
print(len(names), "distinct kernels")
PY
rm -rf $out/trace
