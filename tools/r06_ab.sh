#!/bin/bash
# round 6, on the GPU box: the suite, then the A/B and ablation runs of the round (logs under gpurun_out/)
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06d_gputests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06d_gputests.log
P=raytracinginrust_amd/csrc
echo "=== pair rule A/B (cornell 800x800x256)"; python tools/ab.py --scene cornell --spp 256 --rounds 6 base=$P/librt_amd.so nopair=$P/abx/nopair.so lean4w=$P/abx/lean4w.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_pair_rule_ab.log
echo "=== pair rule A/B (teapot room 800x800x64)"; python tools/ab.py --scene teapot --spp 64 --rounds 5 base=$P/librt_amd.so nopair=$P/abx/nopair.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_pair_rule_ab.log
echo "=== C2 ablations"; python tools/ablate_c2.py --spp 256 base=$P/librt_amd.so onearm=$P/abx/onearm.so lightarm=$P/abx/lightarm.so nopair=$P/abx/nopair.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_ablations.log
echo "=== C2 sections (RT_DIAG build)"; RT_WORKLOADS=C2 python tools/diag_sections.py 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_diag_sections_C2.log
