#!/bin/bash
# tools/prof.sh <tag> [bench args...]  -- run on the GPU box (via gpurun).
# 1) rocprofv3 --kernel-trace --stats (CSV) of bench.py; 2) PMC passes (each in its own run,
# kernel-trace only, as MI355X_MICROARCH.md prescribes).  Output under gpurun_out/prof_<tag>/.
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 40 --warmup 10 --no-cpu-baseline $*"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-400
pmc() { # name counters...
  local name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python bench.py $ARGS > $OUT/bench_pmc_$name.log 2>&1
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
