#!/bin/bash
# tools/prof_mem.sh <tag> <pass> [bench args...] -- ONE memory-path PMC pass for the render kernel (run via gpurun).
# Every pass is wrapped in `timeout`: some counter sets hang rocprofv3 on this pool.
TAG=$1; PASS=$2; shift; shift
OUT=gpurun_out/profmem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline $*"
case $PASS in
  tlb) C="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum";;
  tcp) C="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum";;
  ta)  C="TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum";;
  tcc) C="TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum";;
  ea)  C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE";;
esac
timeout 90 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$PASS -o p -- python bench.py $ARGS > $OUT/bench_$PASS.log 2>&1
echo "pass $PASS rc=$?"
python tools/prof_summary.py $OUT | grep -A8 "PMC pmc_$PASS"
