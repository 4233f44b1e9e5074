#!/bin/bash
# Developer aid: build a copy of the library with another extract_kernels.hip (and extra compiler flags) into stamps/<name>.so
# usage: [VARIANT_FILE=window_kernels.hip] bash tools/build_variant.sh <name> <source file> [flags...]   (default: extract_kernels.hip)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/bv_$name
rm -rf $b && mkdir -p $b/orb_slam2_amd $b/include $root/stamps
cp -r $root/orb_slam2_amd/csrc $b/orb_slam2_amd/ && cp $root/include/* $b/include/ && f=${VARIANT_FILE:-extract_kernels.hip}
cp "$src" $b/orb_slam2_amd/csrc/$f
cd $b/orb_slam2_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $*"
/opt/rocm/bin/hipcc $F -c $f -o ${f%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/stamps/$name.so *.o
/opt/rocm/bin/hipcc $F -S --cuda-device-only -o /tmp/$name.s $f 2>/dev/null
echo built $name: $(grep -A40 "amdhsa_kernel _ZN6orbgpu12k_fast_cellsILi48" /tmp/$name.s | grep -i "next_free_vgpr")
