#!/bin/bash
# A/B builds of the library:  bash tools/build_variant.sh <name> "<extra hipcc flags>" [objects to rebuild with them, default icp_kernels.o]
#   ->  build_variants/<name>/loc_lib_amd/liblocgpu.so     (run with LOCGPU_LIB=<that path>; build_variants/ travels to the GPU box, not into git)
set -e
name=$1; flags=$2; shift; shift
objs=${*:-icp_kernels.o}
root=$(cd "$(dirname "$0")/.." && pwd)
dir=$root/build_variants/$name
rm -rf $dir && mkdir -p $dir/loc_lib_amd $dir/include
cp -rp $root/loc_lib_amd/csrc $dir/loc_lib_amd/csrc
cp -p $root/include/locgpu.h $dir/include/
(cd $dir/loc_lib_amd/csrc && rm -f $objs && make XFLAGS="$flags" 2>&1 | grep -i "error" || true)
rm -f $dir/loc_lib_amd/csrc/*.o
ls -la $dir/loc_lib_amd/liblocgpu.so
