# How much of the STEP a kernel family is worth: timing-only builds that leave its launches out where that keeps the numbers
# the other kernels work on ordinary (zero weight gradients: Adam leaves those weights where they are), same box, interleaved.
#   tools/build_variant.sh wgskip1 az_conv3d_wgrad16.hip -DWG16_SKIP=1   (the six V0 weight gradients)
#   tools/build_variant.sh wgskip2 az_conv3d_wgrad16.hip -DWG16_SKIP=2   (every stride-1 3-D weight gradient of that kernel)
#   tools/build_variant.sh x16a1 az_conv2d_wgrad16.hip -DX16_ABL=1       (2-D 64-channel weight gradients without their flush)
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'loss', d['loss'])"; }
for r in 1 2; do
  echo "shipped   $(run)"
  for v in "$@"; do echo "$v   $(AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_$v.so run)"; done
done
