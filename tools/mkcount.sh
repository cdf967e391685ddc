#!/bin/bash
# EXPERIMENT build for tools/passrate_probe.py: the library with -DRT_COUNT_NODES (box steps count visits / passes per node when
# RT_NODE_COUNTS is set; the flattener takes the contraction set from the file RT_COLLAPSE_SET names) -> csrc/abx/count.so.  Never shipped.
set -e
cd "$(dirname "$0")/../raytracinginrust_amd/csrc"
mkdir -p abx
CXX="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DRT_COUNT_NODES"
/opt/rocm/bin/hipcc $CXX --offload-arch=gfx950 -I. -mllvm -disable-machine-licm -mllvm -enable-misched=0 -DRT_TU=1 -c rt_kernel.hip -o abx/count_lean.o &
/opt/rocm/bin/hipcc $CXX --offload-arch=gfx950 -I. -mllvm -disable-machine-licm -DRT_TU=2 -c rt_kernel.hip -o abx/count_rest.o &
/opt/rocm/bin/hipcc $CXX --offload-arch=gfx950 -x hip -c rt_host.cpp -o abx/count_host.o &
/opt/rocm/bin/hipcc $CXX --offload-arch=gfx950 -x hip -c rt_multi.cpp -o abx/count_multi.o &
/opt/rocm/bin/hipcc $CXX -x c++ -c rt_flatten.cpp -o abx/count_flatten.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abx/count.so abx/count_lean.o abx/count_rest.o abx/count_host.o abx/count_multi.o abx/count_flatten.o rt_jpeg.o rt_obj.o
rm -f abx/count_*.o
echo built abx/count.so
