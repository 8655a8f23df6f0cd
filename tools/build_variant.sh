#!/bin/bash
# Timing-only / experimental variant of ONE csrc file, linked with the current objects of the rest:
#   tools/build_variant.sh <name> <file.hip> "<extra hipcc flags>"   ->  activezero_amd/lib/variants/libazhip_<name>.so
# (run with AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_<name>.so; variants travel with gpurun, not with git)
set -e
name=$1; file=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/activezero_amd/lib
mkdir -p $d/variants/obj_$name
stem=$(basename $file .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c $root/activezero_amd/csrc/$file -o $d/variants/obj_$name/$stem.o
objs=$(ls $d/obj/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/variants/libazhip_$name.so $objs $d/variants/obj_$name/$stem.o
echo $d/variants/libazhip_$name.so
