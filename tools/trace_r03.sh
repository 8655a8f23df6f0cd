set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r03c}; out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --eager-steps 0 > $out/trace_bench.log 2>&1
cd $root
python3 tools/prof_summary.py $out/trace/t_kernel_trace.csv --warmup 2 --out $out/${tag}_bench_b4 --note "${2:-round 3}"
head -40 $out/${tag}_bench_b4_summary.md; python3 tools/stream_timeline.py $out/trace/t_kernel_trace.csv > $out/${tag}_stream_timeline.txt; cat $out/${tag}_stream_timeline.txt
