#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; summary and stream timeline into gpurun_out/<tag>/
#   tools/trace_step.sh <tag> ["note"]
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --eager-steps 0 --no-stage-bench --no-solo-probe > $out/trace_bench.log 2>&1
cd $root
python3 tools/prof_summary.py $out/trace/t_kernel_trace.csv --warmup 2 --out $out/${tag}_bench_b4 --note "$2"
python3 tools/stream_timeline.py $out/trace/t_kernel_trace.csv 3 > $out/${tag}_stream_timeline.txt
rm -f $out/trace/t_kernel_trace.csv  # (tens of MB: the summaries are what is kept)
cat $out/${tag}_stream_timeline.txt
