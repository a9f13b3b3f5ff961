#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/r2_stream.py
OUT=gpurun_out/profmem4; mkdir -p $OUT
timeout 90 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_lat -o p -- python tools/r2_stream.py > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/profmem4/pmc_lat/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'range_check' in r['Kernel_Name']: acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, v in acc.items(): print(k, sum(v.values())/len(v))
PY
