#!/bin/bash
# usage: tools/build_variant.sh <name> [patch.py]  — copies stratum_amd/csrc and include/ to /tmp/variants/<name>/, applies the
# given python patch file (argument 2, optional, run inside the csrc copy), and builds _variants/<name>.so with the product's
# flags plus $EXTRA_FLAGS. Variants may be built in parallel (each has its own tree).
set -e
name=$1; patch=$2
root=$(cd "$(dirname "$0")/.." && pwd)
base=/tmp/variants/$name
d=$base/pkg/csrc
rm -rf $base; mkdir -p $base/pkg; cp -r $root/stratum_amd/csrc $d; cp -r $root/include $base/include
if [ -n "$patch" ]; then (cd $d && python3 $patch); fi
mkdir -p $root/_variants
# one hipcc process per translation unit (as __graft_entry__.build_product), then the link
mkdir -p $base/obj
for f in api.hip shade_plain.hip shade_lt.hip shade_media.hip shade_media_lt.hip shade_media_lt2.hip trace_kernels.hip lbvh.hip lvc.hip hashgrid.hip wide.hip bvh_build.cpp hdr_writer.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-cuda-compat $EXTRA_FLAGS -c -o $base/obj/$f.o $d/$f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_variants/$name.so $base/obj/*.o
echo built _variants/$name.so
