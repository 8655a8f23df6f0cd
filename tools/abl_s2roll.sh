# timing-only ablations of az_conv3d_s2roll.hip (tools/build_variant.sh s2r<N> az_conv3d_s2roll.hip -DS2R_ABL=<N>):
# 1 no output stores, 2 no weight loads after a plane's first tap, 4 no slab staging, 8 no fragment reads
run() { timeout -k 10 120 python tools/bench_s2_family.py --only-s2roll --presplit 2>&1 | grep "m1_32_64"; }
echo "== shipped"; run
for v in "$@"; do
  echo "== abl $v"; AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_s2r$v.so run
done
