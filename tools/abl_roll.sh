# timing-only ablations of az_conv3d_roll.hip (tools/build_variant.sh abl<N> az_conv3d_roll.hip -DR16_ABL=<N>)
run() { timeout -k 10 120 python tools/bench_v0.py 2>&1 | grep "V0 fwd\|V0 dgrad"; }
echo "== shipped"; run
for v in "$@"; do
  echo "== abl $v"; AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_abl$v.so run
done
