#!/bin/bash
# tools/prof_mall.sh <tag>: average L2 -> fabric read latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) of the render kernels of config 5 / config 3 fp32 /
# config 3 next to a MALL-resident and an HBM-resident streaming read (tools/mall_probe.py): how much of FETCH_SIZE's bytes come out of the Infinity Cache.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/mall_${1:-r05}; mkdir -p $OUT
C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum"
run() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$name -o p -- "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run mall64 python tools/mall_probe.py 64 40
run mall128 python tools/mall_probe.py 128 20
run hbm8g python tools/mall_probe.py 8192 4
for wl in cfg5 cfg3_f32 cfg3; do run $wl python bench.py --workload $wl --steps 8 --warmup 2 --profile-clean; done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
print("average L2 -> fabric read latency per kernel (TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum, L2 clocks), requests per launch, share of 32-byte requests, share tagged DRAM")
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files: continue
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if not ("render_band" in k or "stream_probe" in k or "render_lds" in k): continue
        kk = "render_band" if "render_band" in k else "render_lds" if "render_lds" in k else "stream_probe"
        acc[kk][row["Counter_Name"]] += float(row["Counter_Value"]); n[kk].add(row["Dispatch_Id"])
    for kk, c in acc.items():
        rd = c.get("TCC_EA0_RDREQ_sum", 0.0)
        if rd <= 0: continue
        nl = len(n[kk])
        print(f"  {os.path.basename(d.rstrip('/')):10s} {kk:13s} launches {nl:4d}  RDREQ/launch {rd / nl:14.0f}  latency {c.get('TCC_EA0_RDREQ_LEVEL_sum', 0) / rd:8.1f}  "
              f"32B share {c.get('TCC_EA0_RDREQ_32B_sum', 0) / rd:6.3f}  DRAM-tagged share {c.get('TCC_EA0_RDREQ_DRAM_sum', 0) / rd:6.3f}")
for f in sorted(glob.glob(os.path.join(out, "*.log"))):
    for l in open(f):
        if l.startswith("stream probe"): print("  ", os.path.basename(f), l.strip())
PY
