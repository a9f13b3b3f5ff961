"""Collects a tools/prof.sh output directory: kernel stats of the traced bench run, PMC means per dispatch of the render
kernel for every workload, and hbm_traffic.json (FETCH_SIZE x 2: the gfx950 wide-read correction of MI355X_MICROARCH.md,
HBM section; keyed by the kernel-source hash so that bench.py only quotes it for the sources it was measured on)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]


def render_kernel(name):
    return ("render_" in name or "band_table" in name) and "backward" not in name


# issue cost per wave64 VALU instruction and SIMD of each kernel's instruction mix (profiles/r03_probe.txt, r02b_mix_rate_steady.txt):
# fp32 fma / mul / add 1.05-1.1 ns, integer / conversion / floor / compare 1.6-1.9 ns
VALU_NS = {"render_band_kernel": 1.18, "render_lds_kernel": 1.30, "render_wave_kernel": 1.30, "render_dma_kernel": 1.25}


def pmc_means(d):
    """Counter totals of ONE render step = all kernels of the step (AUTO on large bf16 launches: table kernel + band kernel + the tile kernel's
    gated launch), divided by the number of dispatches of the step's dominant kernel."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}, None
    tot = defaultdict(lambda: defaultdict(float))       # counter -> kernel -> total
    disp = defaultdict(set)                             # kernel -> dispatch ids
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if not render_kernel(k):
            continue
        tot[row["Counter_Name"]][k] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    if not tot:
        return {}, None
    any_counter = next(iter(tot.values()))
    kname = max(any_counter, key=any_counter.get)       # the kernel that carries the step
    steps = len(disp[kname])
    return {c: sum(v.values()) / steps for c, v in tot.items()}, kname


for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats of `python bench.py --steps 30 --warmup 5`:")
    for row in csv.DictReader(open(f)):
        print(f"  {row.get('Name', '')[:100]:100s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total%={row.get('Percentage')}")
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    dur = defaultdict(list)
    for row in csv.DictReader(open(tr[0])):
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        if render_kernel(k):
            v2 = v[5:] if len(v) > 5 else v
            print(f"== {k[:110]}: n={len(v)} avg_us(after warm-up)={sum(v2) / len(v2) / 1e3:.1f} min_us={min(v) / 1e3:.1f}")
def bench_line(path):
    if not os.path.isfile(path):
        return ""
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return lines[-1] if lines else ""


line = bench_line(os.path.join(out, "bench_trace.log"))
print("== bench line of the traced run:", line[:600])

from ml_gmpi_amd import _lib  # noqa: E402
traffic = {"_comment": "HBM bytes per launch of the render kernel: rocprofv3 --pmc FETCH_SIZE in its own pass (tools/prof.sh), in KB, "
                       "x 1024 x 2 (gfx950 reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is "
                       "uncalibrated and left out.  valu_insts_per_launch: SQ_INSTS_VALU (wave64 instructions).",
           "source_hash": _lib.source_hash(), "workloads": {}}
for wl in ("cfg3", "cfg2", "cfg3_f32", "cfg4", "cfg5"):
    d = os.path.join(out, wl)
    if not os.path.isdir(d):
        continue
    print(f"== {wl}")
    ent = {"variant": "auto", "source": f"profiles/{os.path.basename(out).replace('prof_', '')}_prof_summary.txt"}
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        means, kname = pmc_means(p)
        if kname:
            ent["kernel"] = kname[:120]
        for c, v in sorted(means.items()):
            extra = ""
            if c == "FETCH_SIZE":
                ent["fetch_size_kb_raw"] = round(v, 1)
                ent["hbm_bytes_per_launch"] = int(v * 1024 * 2)
                extra = f"  -> x2 = {v * 1024 * 2 / 1e9:.3f} GB per launch"
            if c == "SQ_INSTS_VALU":
                ent["valu_insts_per_launch"] = int(v)
                ent["valu_ns_per_inst"] = next((ns for key, ns in VALU_NS.items() if kname and key in kname), 1.30)
            print(f"  {os.path.basename(p):10s} {c:32s} {v:18.1f}{extra}")
    log = os.path.join(d, "bench_fetch.log")
    if os.path.isfile(log):
        try:
            bl = json.loads(bench_line(log))
            ent["algorithmic_bytes_per_launch"] = bl["roofline"]["algorithmic_bytes_per_launch"]
            if "hbm_bytes_per_launch" in ent:
                print(f"  traffic / algorithmic = {ent['hbm_bytes_per_launch'] / ent['algorithmic_bytes_per_launch']:.3f}")
        except Exception as e:
            print("  (no bench line:", e, ")")
    if "hbm_bytes_per_launch" in ent:
        traffic["workloads"][wl] = ent
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print("== wrote", os.path.join(out, "hbm_traffic.json"), "for sources", traffic["source_hash"])
