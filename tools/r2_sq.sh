#!/bin/bash
# issue-slot counters of one render variant at fitting poses: tools/r2_sq.sh <tag> <dtype> <variant> <tune>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; DT=$2; VAR=$3; TUNE=$4
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
pass() { local name=$1; shift
  GMPI_TUNE_WAVE=$TUNE timeout 100 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python tools/r2_pose.py $DT $VAR short > $OUT/run_$name.log 2>&1
  echo "== $TAG $name rc=$?"; python tools/prof_summary.py $OUT | grep -A9 "PMC pmc_$name" | tail -n +2; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
