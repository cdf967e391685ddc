#!/bin/bash
# Build a kernel variant as csrc/abx/<name>.so for A/B timing with tools/ab.py (built here and shipped to the GPU box with the snapshot; *.so is git-ignored).
# usage: tools/mkab.sh <name> "<extra flags for the lean TU>" "<extra flags for the other TU>" [kernel source (default rt_kernel.hip)]
set -e
cd "$(dirname "$0")/../raytracinginrust_amd/csrc"
name=$1; f1=$2; f2=$3; src=${4:-rt_kernel.hip}
mkdir -p abx
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 -I. -mllvm -disable-machine-licm"   # as the Makefile
/opt/rocm/bin/hipcc $BASE -mllvm -enable-misched=0 $f1 -DRT_TU=1 -c $src -o abx/${name}_lean.o &
/opt/rocm/bin/hipcc $BASE $f2 -DRT_TU=2 -c $src -o abx/${name}_rest.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abx/$name.so abx/${name}_lean.o abx/${name}_rest.o rt_host.o rt_multi.o rt_flatten.o rt_jpeg.o rt_obj.o
rm -f abx/${name}_lean.o abx/${name}_rest.o
echo built abx/$name.so
