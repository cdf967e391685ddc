#!/bin/bash
# Runs on the GPU box: kernel-trace stats + the PMC passes for each workload given.  usage: tools/profile_all.sh <tag> C1 C3 ...
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  O=/root/repo/gpurun_out/prof_${TAG}_$W
  mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-spp 0 --also none --workload $W > $O/bench.json 2> $O/bench.err
  tail -c 400 $O/bench.json; echo
  bash /root/repo/tools/profile_pmc.sh ${TAG}_$W --workload $W > $O/pmc.log 2>&1
  cat /root/repo/gpurun_out/pmc_${TAG}_$W/summary.csv
done
