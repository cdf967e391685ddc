#!/bin/bash
# usage: ab_run.sh "<variants name=path ...>"
V="$@"
for sc in "cornell 256" "random 256" "final 64" "teapot 64"; do set -- $sc; echo "=== $1 (spp $2)"; python tools/ab.py --scene $1 --spp $2 --rounds 4 $V 2>&1 | grep -v amdgpu.ids; done
