#!/bin/bash
# tools/pose_table.sh <out dir>  -- run on the GPU box: tile (lds) vs strip (wave) vs AUTO over small launches and camera yaws
# (sets p<S>_y<yaw> of tools/kbench_dump.py: all views at the same yaw, pitch 0; kbench = C ABI + HIP events at steady clocks).
cd ${GRAFT_REPO_ROOT:-.}
O=${1:-gpurun_out/pose}; mkdir -p $O
K=tools/ubench/bin/kbench
L=ml-gmpi_amd/libgmpi_render.so
: > $O/raw.txt
for dt in f32 bf16; do
  for S in 256 512; do
    if [ $S = 256 ]; then VS="1 2 4 8"; else VS="1 2"; fi
    for y in 000 300 450 578; do
      for v in $VS; do
        echo "== $dt p${S}_y$y views $v" >> $O/raw.txt
        timeout 60 $K $L p${S}_y$y $dt lds,wave,auto 10 $v 2>&1 | grep mean >> $O/raw.txt
      done
    done
  done
done
python3 - $O <<'PY'
import re, sys
rows, cur = [], None
for l in open(sys.argv[1] + "/raw.txt"):
    if l.startswith("=="):
        cur = l.split()[1:]; res = {}; rows.append((cur, res))
    else:
        m = re.search(r"\s(lds|wave|auto)\s+mean\s+([\d.]+) ms", l)
        if m: res[m.group(1)] = float(m.group(2))
out = open(sys.argv[1] + "/table.txt", "w")
hdr = f"{'dtype':5s} {'set':10s} {'views':>5s} {'tile ms':>9s} {'strip ms':>9s} {'auto ms':>9s} {'auto/best':>9s}"
print(hdr); out.write(hdr + "\n")
worst = 0
for (dt, st, _, v), r in rows:
    if len(r) < 3: continue
    best = min(r["lds"], r["wave"]); ratio = r["auto"] / best; worst = max(worst, ratio)
    line = f"{dt:5s} {st:10s} {v:>5s} {r['lds']:9.4f} {r['wave']:9.4f} {r['auto']:9.4f} {ratio:9.3f}"
    print(line); out.write(line + "\n")
out.write(f"worst auto/best = {worst:.3f}\n"); print(f"worst auto/best = {worst:.3f}")
PY
