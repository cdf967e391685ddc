#!/bin/bash
# Copies what tools/profile_all.sh left under gpurun_out/ (scratch) into profiles/ (tracked) under the names the README documents.
# usage: tools/collect_profiles.sh <tag> C1 C2 ...
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
for W in "$@"; do
  tail -1 gpurun_out/prof_${TAG}_$W/bench.json > profiles/${TAG}_bench_$W.json
  cp "$(ls -t gpurun_out/prof_${TAG}_$W/trace/*/*_kernel_stats.csv | head -n 1)" profiles/${TAG}_bench_${W}_kernel_stats.csv   # (the newest: gpurun_out keeps earlier runs)
  cp gpurun_out/pmc_${TAG}_$W/summary.csv profiles/${TAG}_bench_${W}_pmc_summary.csv
  [ -f gpurun_out/prof_${TAG}_$W/loop.json ] && cp gpurun_out/prof_${TAG}_$W/loop.json profiles/${TAG}_bench_${W}_loop.json   # which loop shape the un-profiled call found (and the profiled ones were told to run)
done
ls -la profiles/${TAG}_*
