#!/bin/bash
# Issue-slot counters of one kernel configuration through the torch-free harness (seconds per pass on the GPU box):
#   tools/kprof.sh <tag> <lib.so> <set> <dtype> <variant> <GMPI_TUNE_WAVE>   ->  gpurun_out/kp_<tag>/summary.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=$1; LIB=$2; SET=$3; DT=$4; VAR=$5; TUNE=${6:-0}
OUT=gpurun_out/kp_$TAG
mkdir -p $OUT
pass() { local name=$1; shift
  GMPI_TUNE_WAVE=$TUNE timeout 100 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- tools/ubench/bin/kbench $LIB $SET $DT $VAR 6 > $OUT/run_$name.log 2>&1
  echo "== $TAG $name rc=$?"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
pass c TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass d FETCH_SIZE
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files: print(d, "no counters"); continue
    acc = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(files[0])):
        if "render_" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for c, v in sorted(acc.items()):
        print(f"  {os.path.basename(d):6s} {c:34s} {sum(v.values()) / len(v):16.1f}")
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if "render_" in r["Kernel_Name"]]
        if dur: print(f"  {os.path.basename(d):6s} kernel avg_us {sum(dur) / len(dur) / 1e3:.1f} min_us {min(dur) / 1e3:.1f} n {len(dur)}")
PY
