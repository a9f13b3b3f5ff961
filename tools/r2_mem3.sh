#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pass() { local tag=$1 var=$2 tune=$3 name=$4; shift 4
  local OUT=gpurun_out/profmem3_$tag; mkdir -p $OUT
  GMPI_TUNE_WAVE=$tune timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python tools/r2_pose.py bf16 $var short > $OUT/run_$name.log 2>&1
  echo "== $tag $name rc=$?"; python tools/prof_summary.py $OUT | grep -A6 "PMC pmc_$name" | tail -n +2; }
for cfg in "wave512 wave 512" "lds lds 0"; do set -- $cfg
  pass $1 $2 $3 lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pass $1 $2 $3 tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
