#!/bin/bash
# memory-path counters, one set per run (each in its own rocprofv3 pass, kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pass() { # tag variant tune name counters...
  local tag=$1 var=$2 tune=$3 name=$4; shift 4
  local OUT=gpurun_out/profmem_$tag
  mkdir -p $OUT
  GMPI_TUNE_WAVE=$tune timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --workload cfg3_f32 --variant $var > $OUT/bench_$name.log 2>&1
  echo "== $tag $name rc=$?"
  python tools/prof_summary.py $OUT | grep -A6 "PMC pmc_$name" | tail -n +2
}
for cfg in "wave512 wave 512" "lds lds 0" "wave0 wave 0"; do
  set -- $cfg
  pass $1 $2 $3 lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  pass $1 $2 $3 ta TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum
  pass $1 $2 $3 tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass $1 $2 $3 tcp2 TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum
done
