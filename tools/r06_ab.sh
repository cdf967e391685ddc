#!/bin/bash
# round 6, on the GPU box: the suite, then the A/B runs of the round (logs under gpurun_out/)
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06e_gputests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06e_gputests.log
P=raytracinginrust_amd/csrc
for sc in "cornell 256" "random 256" "final 64" "teapot 64"; do set -- $sc
  V="base=$P/librt_amd.so flush2=$P/abx/flush2.so"; [ $1 = cornell ] && V="$V flipfast=$P/abx/flipfast.so"
  echo "=== per-lane atomics at the accumulator flush: $1 (spp $2)"; python tools/ab.py --scene $1 --spp $2 --rounds 5 $V 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_flush_per_lane_ab.log
python tests/sweeps/full_frame_sweep.py C5 --minutes 9 --out gpurun_out/r06e_full_frame_C5.json 2>&1 | tail -2
