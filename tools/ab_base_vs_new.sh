set -o pipefail
B=$PWD/activezero_amd/lib/variants/libazhip_base.so
python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_psmnet.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; tail -3 gpurun_out/ab_tests.log
for i in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export AZ_LIB_PATH=$B; else unset AZ_LIB_PATH; fi
    echo "== $v $i"
    python tools/bench_v0.py 2>&1 | grep "V0 fwd"
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --eager-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', round(d['ms_per_step'],2))"
  done
done
