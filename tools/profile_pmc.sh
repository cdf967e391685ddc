#!/bin/bash
# Runs on the GPU box: PMC passes for the bench workload (each pass = its own rocprofv3 run; no trace domains
# are combined with --pmc).  Usage: tools/profile_pmc.sh <tag> [bench args...]   (for a mesh workload pass --loop persistent|lockstep:
# tools/profile_all.sh does, after asking an un-profiled call which one the scene runs)
set -u
TAG=${1:-r01}; shift || true
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --cpu-spp 0 --also none "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
  tail -1 $OUT/$name.log | cut -c1-200
}
BENCH_ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 /root/repo/tools/pmc_summary.py $OUT
