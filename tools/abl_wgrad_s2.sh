# timing-only ablations of az_conv3d_wgrad16s2.hip (tools/build_variant.sh s2a<N> az_conv3d_wgrad16s2.hip -DS2_ABL=<N>)
run() { timeout -k 10 120 python tools/bench_s2_family.py --only-s2-wgrad 2>&1 | grep "wgrad_s2"; }
echo "== shipped"; run
for v in "$@"; do
  echo "== abl $v"; AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_s2a$v.so run
done
