#!/bin/bash
# tools/prof.sh <tag>  -- run on the GPU box (via gpurun): rocprofv3 evidence for the bench workloads.
#   1) kernel trace + stats of the default bench command (cfg3)                      -> $OUT/trace
#   2) per workload, each in its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, HBM section):
#      FETCH_SIZE pass and SQ_INSTS_VALU pass                                          -> $OUT/<workload>/pmc_*
#   3) issue-slot counters of the headline kernel (cfg3), two passes                   -> $OUT/cfg3/pmc_sq*
#   4) tools/prof_collect.py: summary.txt + hbm_traffic.json keyed by the kernel-source hash (bench.py quotes it only
#      for matching sources).  Copy both into profiles/ afterwards.
set -u
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
# --profile-clean: the render kernels are launched W + K times and never else (clock ramp on the stream-probe kernel; no parity, pose sweep, end-to-end
# loop, companions, CPU baseline): the kernel-trace averages ARE the timed steps, the counter means are per timed launch
ARGS="--steps 12 --warmup 4 --profile-clean"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --steps 30 --warmup 5 --profile-clean > $OUT/bench_trace.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-300
pmc() { # workload name counters...
  local wl=$1 name=$2; shift 2
  mkdir -p $OUT/$wl
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$wl/pmc_$name -o p -- python bench.py $ARGS --workload $wl > $OUT/$wl/bench_$name.log 2>&1
  echo "$wl $name rc=$?"
}
for wl in ${WORKLOADS:-cfg3 cfg2 cfg3_f32 cfg4 cfg5}; do
  pmc $wl fetch FETCH_SIZE
  pmc $wl valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
done
pmc cfg3 sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc cfg3 sq2 SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU
pmc cfg3 tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc cfg3 tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
# the G-step shapes (forward + backward): trace + FETCH / WRITE / issue counters of both kernels
for wl in ${TRAIN:-train256 train1024}; do
  mkdir -p $OUT/$wl
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl/trace -o t -- python bench.py --workload $wl --steps 20 --warmup 5 --no-parity > $OUT/$wl/bench_trace.log 2>&1
  pmc_t() { local name=$1; shift
    timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$wl/pmc_$name -o p -- python bench.py --workload $wl --steps 8 --warmup 2 --prewarm-ms 0 --no-parity > $OUT/$wl/bench_$name.log 2>&1
    echo "$wl $name rc=$?"; }
  pmc_t fetch FETCH_SIZE
  pmc_t write WRITE_SIZE
  pmc_t valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc_t sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
done
python tools/prof_collect.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
