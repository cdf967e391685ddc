#!/bin/bash
# Like tools/mkab.sh for a variant of the LEAN translation unit only (the list-scene kernels: C2 / C5): compiles that unit from `src` with
# extra flags and links it with the product build's other objects -> csrc/abx/<name>.so.  ~40 s instead of ~3 min.
# usage: tools/mkab_lean.sh <name> "<extra flags>" [kernel source (default rt_kernel.hip)]
set -e
cd "$(dirname "$0")/../raytracinginrust_amd/csrc"
name=$1; f1=$2; src=${3:-rt_kernel.hip}
mkdir -p abx
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-pass-failed --offload-arch=gfx950 -I. -mllvm -disable-machine-licm"   # as the Makefile
/opt/rocm/bin/hipcc $BASE -mllvm -enable-misched=0 $f1 -DRT_TU=1 -c $src -o abx/${name}_lean.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abx/$name.so abx/${name}_lean.o rt_kernel_rest.o rt_host.o rt_multi.o rt_flatten.o rt_jpeg.o rt_obj.o
rm -f abx/${name}_lean.o
echo built abx/$name.so
