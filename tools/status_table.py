"""The status tables of DESIGN.md section 7 and README.md, printed from the round's evidence files (profiles/r<NN>_bench_lines.jsonl, r<NN>_default_bench.json,
r<NN>_train_lines.jsonl, r<NN>_video_bench.json): one source for the numbers the two documents quote.    usage: python tools/status_table.py [r06]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def lines(name):
    with open(os.path.join(ROOT, "profiles", f"{tag}_{name}")) as f:
        return [json.loads(l) for l in f if l.startswith("{")]


bench = lines("bench_lines.jsonl")
default = lines("default_bench.json")[-1]
train = lines("train_lines.jsonl")
video = lines("video_bench.json")[-1]
comp = default["companions"]
comp_of = {("cfg3", False): None, ("cfg3_f32", False): "cfg3_f32", ("cfg5", False): "cfg5_shard", ("cfg2", False): "cfg2", ("cfg4", False): "cfg4_shard", ("cfg3", True): "cfg3_strict"}


def rng(v):
    return f"{min(v)}–{max(v)}" if min(v) != max(v) else f"{v[0]}"


print(f"sources {json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))['source_hash']}")
print("| workload | ms per step (companion) | frac (companion) | frac_footprint | frac_pose_mean (mean / p90 / worst ms) | traffic / algorithmic | MHz, W |")
for d in bench:
    c, r = d["config"], d["roofline"]
    key = (c["name"], bool(c.get("strict_order")))
    co = default if comp_of[key] is None else comp[comp_of[key]]
    co_ms = co["ms_per_step"] if comp_of[key] is None else co["ms"]
    co_frac = co["roofline"]["frac"] if comp_of[key] is None else co["frac"]
    sw = d.get("pose_sweep")
    pose = "—" if not sw else f"{r['frac_pose_mean']:.3f} ({sw['mean_ms']:.3f} / {sw['p90_ms']:.3f} / {sw['worst_ms']:.3f}; off band {sw['views_off_band_share']})"
    tr = "—" if not r.get("traffic") else f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.3f}"
    pw = d["power"]
    print(f"| {c['name']}{' strict' if key[1] else ''} | {d['ms_per_step']:.3f} ({co_ms:.3f}) | {r['frac']:.3f} ({co_frac:.3f}) | {r['frac_footprint']:.3f} | {pose} | {tr} | "
          f"{rng(pw['engine_mhz'])} MHz, {rng([int(w) for w in pw['package_w']])} W |")
print("G-step:", " / ".join(f"{d['ms_per_step']:.3f}" for d in train), "= forward", " / ".join(f"{d['roofline']['parts']['forward_ms']:.3f}" for d in train),
      "+ zero-fill", " / ".join(f"{d['roofline']['parts']['grad_zero_fill_ms']:.3f}" for d in train),
      "+ backward", " / ".join(f"{d['roofline']['parts']['backward_ms']:.3f}" for d in train),
      "(gather pair", " / ".join(f"{d['roofline']['parts']['backward_gather_ms']:.3f}" for d in train) + ")",
      "| frac", " / ".join(f"{d['roofline']['frac']:.2f}" for d in train), "backward frac", " / ".join(f"{d['roofline']['parts']['backward_frac']:.2f}" for d in train),
      "| traffic", " / ".join("—" if not d["roofline"].get("traffic") else f"{d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch']:.2f}" for d in train),
      f"| companions.train1024 {comp['train1024']['ms']:.3f} ms")
c3 = bench[0]
print("host, config 3:", {k: v for k, v in c3.items() if k.startswith("e2e")}, "kernel_ms", c3["roofline"]["kernel_ms"])
print("video:", video["views_per_s"], "frames/s per-view loop;", video["batched_driver"]["views_per_s"], "views/s ViewBatchDriver.render_path")
cb = default["cpu_baseline"]
print("cpu:", cb["value"], cb["unit"], "at", cb["cores"], "threads (reference-ops); port", cb["port"]["value"], "on", cb["port"]["cores"])
