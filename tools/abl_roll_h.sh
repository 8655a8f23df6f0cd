# timing-only ablations of the f16x3 stage of az_conv3d_roll.hip (tools/build_variant.sh h<N> az_conv3d_roll.hip -DR16H_ABL=<N>)
run() { timeout -k 10 120 python tools/f16x3_probe.py --time-only 2>&1 | grep "V0 dgrad f16x3"; }
echo "== shipped"; run
for v in "$@"; do
  echo "== abl $v"; AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_h$v.so run
done
