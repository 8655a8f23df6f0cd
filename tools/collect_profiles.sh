#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for a round tag:
#   tools/collect_profiles.sh r01g
# (1) PMC passes (FETCH_SIZE, WRITE_SIZE; --pmc only with --kernel-trace) over tools/kernel_probe.py,
# (2) rocprofv3 --kernel-trace --stats of bench.py, (3) a default bench.py line.
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "pmc fetch" >> $out/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o fetch -- python3 $root/tools/kernel_probe.py > $out/pmc_fetch.log 2>&1
echo "pmc write" >> $out/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc -o write -- python3 $root/tools/kernel_probe.py > $out/pmc_write.log 2>&1
echo "trace" >> $out/progress.txt
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/trace_bench.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/pmc --out $out/pmc_traffic_b4
python3 tools/prof_summary.py $out/trace/t_kernel_trace.csv --warmup 2 --out $out/${tag}_bench_b4 --note "$2"
cp $out/trace/t_kernel_stats.csv $out/${tag}_rocprof_kernel_stats_whole_process.csv 2>/dev/null || true
echo "bench" >> $out/progress.txt
python3 bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
tail -c 600 $out/${tag}_bench_default.json
