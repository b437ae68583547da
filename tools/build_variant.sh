#!/bin/bash
# usage: tools/build_variant.sh <name> [sed-script ...]  — copies stratum_amd/csrc to /tmp/variants/x/y/<name>, applies the
# given python patch file (argument 2, optional) there, and builds _variants/<name>.so with the product's flags.
set -e
name=$1; patch=$2
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/variants/x/y/$name
rm -rf $d; mkdir -p /tmp/variants/x/y; cp -r $root/stratum_amd/csrc $d
rm -rf /tmp/variants/x/include; cp -r $root/include /tmp/variants/x/include
if [ -n "$patch" ]; then (cd $d && python3 $patch); fi
mkdir -p $root/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $EXTRA_FLAGS -o $root/_variants/$name.so $d/api.hip $d/lbvh.hip $d/lvc.hip $d/bvh_build.cpp $d/hdr_writer.cpp
echo built _variants/$name.so
