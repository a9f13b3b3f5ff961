#!/bin/bash
# Round 6, VERDICT r5 item 1: the band kernel's in-view band order as a parameter -- per order {ms, L2 -> fabric bytes}, same box, alternating repeats.
#   tools/band_order.sh <tag>   (on the GPU box; needs ml-gmpi_amd/libgmpi_render_tune.so = tools/build_tune.sh, tools/ubench/bin/kbench, gpurun_in/kb_*.bin)
# GMPI_TUNE_ORDER = band columns per XCD window: 1 = column-major (round 5), 8 / 4 >= bands_x = row-major (rounds 3-4).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/order_$TAG
mkdir -p $OUT
LIB=ml-gmpi_amd/libgmpi_render_tune.so
KB=tools/ubench/bin/kbench
ORDERS=${ORDERS:-"1 2 4 8"}
run() { # set dtype order
  GMPI_TUNE_ORDER=$3 timeout 120 $KB $LIB $1 $2 auto 20 | grep mean | sed "s/^/order $3: /"
}
for rep in 1 2 3; do
  for wl in "bench bf16" "bench f32" "c5 f32"; do
    for o in $ORDERS; do run $wl $o; done
  done
done | tee $OUT/ms.txt
fetch() { # set dtype order
  local d=$OUT/pmc_$1_$2_o$3
  GMPI_TUNE_ORDER=$3 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -o p -- $KB $LIB $1 $2 auto 6 > $d.log 2>&1
  python3 - $d "$1 $2 order $3" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, tag = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files: print(tag, "no counters"); sys.exit()
acc = defaultdict(lambda: defaultdict(float))
for row in csv.DictReader(open(files[0])):
    if "render_band" not in row["Kernel_Name"]: continue
    acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for c, v in sorted(acc.items()):
    print(f"{tag}: {c} mean per band-kernel launch {sum(v.values()) / len(v):.1f} KB raw -> x2 = {sum(v.values()) / len(v) * 2048 / 1e9:.4f} GB  (n {len(v)})")
PY
}
for wl in "bench bf16" "bench f32" "c5 f32"; do
  for o in $ORDERS; do fetch $wl $o; done
done | tee $OUT/fetch.txt
# channel-camping hypothesis: per-instance L2 -> fabric read requests and busy cycles, row-major against column-major (json keeps the instances)
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for o in 1 8; do
  for c in "TCC_EA0_RDREQ TCC_BUSY" "TCC_TAG_STALL TCC_REQ"; do
    n=$(echo $c | tr ' ' '_')
    GMPI_TUNE_ORDER=$o timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format json -d $OUT/chan_o${o}_$n -o p -- $KB $LIB bench f32 auto 2 > $OUT/chan_o${o}_$n.log 2>&1
    echo "chan order $o $c rc=$?"
  done
done
find $OUT -name "*.json" -size +15M -delete; du -sh $OUT
