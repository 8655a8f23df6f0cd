set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r03a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --eager-steps 0 > $out/trace_bench.log 2>&1
cd $root
python3 tools/prof_summary.py $out/trace/t_kernel_trace.csv --warmup 2 --out $out/r03a_bench_b4 --note "round 3: depth-rolling 16x16x32 V0 kernel, fine-row-walk wgrad, K6 backward without LDS atomics"
head -60 $out/r03a_bench_b4_summary.md
