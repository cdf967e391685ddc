#!/bin/bash
# Runs on the GPU box: kernel-trace stats + the PMC passes for each workload given.  usage: tools/profile_all.sh <tag> C1 C3 ...
# Per workload: (1) one un-profiled call finds out which loop shape the scene runs (mesh scenes measure it: rt_scene_calibrate) — the
# profiled calls then SET that shape (--loop), so that no calibration launch lands in a per-kernel average or a counter summary;
# (2) rocprofv3 --kernel-trace --stats; (3) the counter passes (tools/profile_pmc.sh), summarised over the timed frames of the
# instantiation the bench line names (tools/pmc_summary.py).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  O=/root/repo/gpurun_out/prof_${TAG}_$W
  mkdir -p $O
  python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-spp 0 --also none --workload $W > $O/probe.json 2> $O/probe.err
  LOOP=$(python3 -c "import json,sys; d=json.loads(open('$O/probe.json').read().strip().splitlines()[-1]); print({'persistent':'persistent'}.get(d['loop']['shape'],'lockstep')); print(json.dumps(d['loop']), file=sys.stderr)" 2> $O/loop.json)
  echo "$W: loop $(cat $O/loop.json) -> --loop $LOOP"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-spp 0 --also none --workload $W --loop $LOOP > $O/bench.json 2> $O/bench.err
  tail -c 400 $O/bench.json; echo
  bash /root/repo/tools/profile_pmc.sh ${TAG}_$W --workload $W --loop $LOOP > $O/pmc.log 2>&1
  tail -3 $O/pmc.log
  cat /root/repo/gpurun_out/pmc_${TAG}_$W/summary.csv
done
