# scheduling A/B runs of the default bench line on ONE box (ms per step)
run() { timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --eager-steps 0 --no-stage-bench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step' % d['ms_per_step'])"; }
echo "== default"; run
echo "== step on a high-priority stream"; AZ_BENCH_HP=1 run
echo "== V0 weight gradient: 256 workgroups (one per CU)"; AZ_WGRAD_R16_WGS=256 run
echo "== both"; AZ_BENCH_HP=1 AZ_WGRAD_R16_WGS=256 run
echo "== high priority + 128 workgroups"; AZ_BENCH_HP=1 AZ_WGRAD_R16_WGS=128 run
echo "== no overlap"; timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --eager-steps 0 --no-stage-bench --no-wgrad-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step' % d['ms_per_step'])"
echo "== default again"; run
echo "== 2-D weight gradients in order on the main stream (3-D ones still on the side stream)"; AZ_2D_WGRAD_OVERLAP=0 run
echo "== default once more"; run
