#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAG=1 ..."  -> nerf_loc_amd/csrc/variants/libnerfloc_NAME.so: the library with ONE translation unit rebuilt with extra flags
# (A/B experiments on the GPU box: NERFLOC_LIB=nerf_loc_amd/csrc/variants/libnerfloc_NAME.so python tools/front_time.py; variants/ is git-ignored)
set -e
cd "$(dirname "$0")/../nerf_loc_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p variants/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function $flags -c $src -o variants/obj_$name/${src%.hip}.o
objs=""
for o in build/*.o; do b=$(basename $o); if [ "$b" == "${src%.hip}.o" ]; then objs="$objs variants/obj_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libnerfloc_$name.so $objs
echo built variants/libnerfloc_$name.so
