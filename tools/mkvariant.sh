#!/bin/bash
# Build a patched copy of the kernel as csrc/variants/<name>.so (ablation / A-B experiments; never shipped).
# usage: tools/mkvariant.sh <name> [-D...] [-- 'old text' 'new text' ...]
set -e
cd "$(dirname "$0")/../raytracinginrust_amd/csrc"
name=$1; shift
defs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do defs+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
mkdir -p variants/src
python3 - "variants/src/$name.hip" "$@" <<'PY'
import sys
dst = sys.argv[1]; s = open('rt_kernel.hip').read()
for i in range(2, len(sys.argv), 2):
    old, new = sys.argv[i], sys.argv[i + 1]
    assert old in s, 'not found: ' + old
    s = s.replace(old, new)
open(dst, 'w').write(s)
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I. -I../../include "${defs[@]}" -c variants/src/$name.hip -o variants/src/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so variants/src/$name.o rt_host.o rt_flatten.o rt_jpeg.o
echo built variants/$name.so
