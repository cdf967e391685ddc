#!/bin/bash
# round 6, on the GPU box: the suite, then the A/B runs of the round (logs under gpurun_out/)
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06f_gputests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06f_gputests.log
P=raytracinginrust_amd/csrc
for sc in "cornell 256" "random 256" "final 64" "teapot 64"; do set -- $sc
  echo "=== this build (Cube leaves from the node's box; FlipNormal-only chains on the path's own ray; per-lane flush in the persistent kernels) against the build before: $1 (spp $2)"
  python tools/ab.py --scene $1 --spp $2 --rounds 6 base=$P/librt_amd.so prev=$P/abx/prev.so 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_cube_leaf_flip_flush_ab.log
