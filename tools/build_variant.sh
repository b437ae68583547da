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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $EXTRA_FLAGS -o $root/_variants/$name.so $d/api.hip $d/lbvh.hip $d/lvc.hip $d/hashgrid.hip $d/wide.hip $d/bvh_build.cpp $d/hdr_writer.cpp
echo built _variants/$name.so
