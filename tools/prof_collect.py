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


TRAIN = False


def render_kernel(name):
    if TRAIN:   # the G-step workloads: forward + backward kernels of the step (the zero-fill is torch's)
        return "render_" in name or "band_table" in name
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
def bench_line(path):
    if not os.path.isfile(path):
        return ""
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return lines[-1] if lines else ""


line = bench_line(os.path.join(out, "bench_trace.log"))
print("== bench line of the traced run:", line[:600])
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    # --profile-clean: every launch of a render kernel in this trace is a warm-up or a timed step of the bench line above, so the step time can
    # be recomputed from this CSV alone: sum over the step's kernels of their average duration
    dur = defaultdict(list)
    for row in csv.DictReader(open(tr[0])):
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    step_us = 0.0
    for k, v in dur.items():
        if render_kernel(k):
            step_us += sum(v) / len(v) / 1e3
            print(f"== {k[:110]}: n={len(v)} avg_us={sum(v) / len(v) / 1e3:.1f} min_us={min(v) / 1e3:.1f} max_us={max(v) / 1e3:.1f}")
    try:
        bl = json.loads(line)
        ab, kms = bl["roofline"]["algorithmic_bytes_per_launch"], bl["roofline"]["kernel_ms"]
        print(f"== step recomputed from the trace: sum of the kernels' averages = {step_us:.1f} us; bench line: kernel_ms {kms * 1e3:.1f} us "
              f"(events around the step's launches), ms_per_step {bl['ms_per_step'] * 1e3:.1f} us; ratio trace / kernel_ms = {step_us / (kms * 1e3):.4f}")
        print(f"== roofline fraction from the trace alone: {ab} B / {step_us:.1f} us / 8 TB/s = {ab / (step_us * 1e-6) / 8e12:.4f}  (line: {bl['roofline']['frac']})")
    except Exception as e:  # noqa: BLE001
        print("  (no bench line to compare with:", e, ")")

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
TRAIN = True
for wl in ("train256", "train512", "train1024"):
    d = os.path.join(out, wl)
    if not os.path.isdir(d):
        continue
    print(f"== {wl} (forward + backward kernels of a step; FETCH_SIZE x 2 as for the render, WRITE_SIZE raw x 1024: uncalibrated)")
    print("   bench line:", bench_line(os.path.join(d, "bench_trace.log"))[:900])
    ent = {"variant": "auto", "source": f"profiles/{os.path.basename(out).replace('prof_', '')}_prof_summary.txt"}
    trw = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for f in trw:
        for row in csv.DictReader(open(f)):
            if float(row.get("Percentage") or 0) > 1.0:
                print(f"   {row.get('Name', '')[:100]:100s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total%={row.get('Percentage')}")
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        files = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        tot = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for row in csv.DictReader(open(files[0])):
            k = row["Kernel_Name"]
            if not render_kernel(k):
                continue
            kk = "backward" if "backward" in k else "forward"
            tot[row["Counter_Name"]][kk] += float(row["Counter_Value"])
            disp[kk].add(row["Dispatch_Id"])
        for c, v in sorted(tot.items()):
            for kk, t in sorted(v.items()):
                per = t / max(len(disp["backward"]), 1)   # (one backward launch per step; a forward step may be several kernels)
                print(f"   {os.path.basename(p):10s} {c:28s} {kk:9s} per step {per:18.1f}")
                if c == "FETCH_SIZE":
                    ent[f"{kk}_fetch_bytes"] = int(per * 1024 * 2)
                if c == "WRITE_SIZE":
                    ent[f"{kk}_write_bytes_uncalibrated"] = int(per * 1024)
    if "forward_fetch_bytes" in ent and "backward_fetch_bytes" in ent:
        ent["hbm_bytes_per_launch"] = ent["forward_fetch_bytes"] + ent["backward_fetch_bytes"] + ent.get("forward_write_bytes_uncalibrated", 0) + ent.get("backward_write_bytes_uncalibrated", 0)
        traffic["workloads"][wl] = ent
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print("== wrote", os.path.join(out, "hbm_traffic.json"), "for sources", traffic["source_hash"])
